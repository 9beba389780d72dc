#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/abk.sh tag "<defines A>" "<defines B>" ... [-- bench args]
# For each set of -D switches: rebuild the kernels on the GPU box, run a short bench under rocprofv3 --kernel-trace
# and print the frame rate, the stage times and the integrate kernels' average durations.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
tag=$1; shift
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
BARGS="--steps 40 --warmup 5"
i=0
for defs in "$@"; do
  make -s -C housescan_amd/csrc FLAGS="$BASE $defs" 2>&1 | grep -E "error" 
  OUT=$ROOT/gpurun_out/$tag/v$i
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --allow-exp --quick $BARGS ${HSK_BENCH_ARGS} > $OUT/log.txt 2>&1)
  echo "== [$i] $defs"
  grep -o '{"metric.*' $OUT/log.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   fps', d['value'], {k:v for k,v in d['stage_us'].items() if k!='note'}, 'frac', d['roofline']['frac'])"
  python3 tools/kstats.py $OUT/trace 14 | grep -i "integrate\|zrange\|raycast\|icp_iter" | sed 's/^/   /'
  rm -rf $OUT/trace
  i=$((i+1))
done
