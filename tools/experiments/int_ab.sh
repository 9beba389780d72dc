#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/int_ab.sh tag "<defines A>" "<defines B>" ...   (env VOL=512|1024)
# integrate-only A/B: rebuild with each set of -D switches, run tools/int_bench.py under rocprofv3, print kernel averages
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
tag=$1; shift
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
i=0
for defs in "$@"; do
  make -s -C housescan_amd/csrc FLAGS="$BASE $defs" 2>&1 | grep -E "error"
  OUT=$ROOT/gpurun_out/$tag/v$i
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/int_bench.py ${VOL:-512} 0 26 > $OUT/log.txt 2>&1)
  echo "== [$i] $defs"
  python3 tools/kstats.py $OUT/trace 14 | grep -i "integrate\|zrange" | grep -v "<true>" | sed 's/^/   /'
  rm -rf $OUT/trace
  i=$((i+1))
done
