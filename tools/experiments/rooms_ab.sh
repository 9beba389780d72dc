#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/rooms_ab.sh REPS "<defines A>" "<defines B>" ... -- as ab_repeat.sh, for the concurrent-room figures (2 and 4 rooms on one GPU)
cd ${GRAFT_REPO_ROOT:-.}
reps=$1; shift
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
i=0
for defs in "$@"; do
  make -s -j8 -C housescan_amd/csrc FLAGS="$BASE $defs" 2>&1 | grep -E "error"
  cp housescan_amd/libhskinfu.so /tmp/libhsk_v$i.so
  echo "v$i = [$defs]"
  i=$((i+1))
done
n=$i
for r in $(seq 1 $reps); do
  for i in $(seq 0 $((n-1))); do
    cp /tmp/libhsk_v$i.so housescan_amd/libhskinfu.so
    python bench.py --allow-exp --no-cpu-baseline --no-traffic --no-1024 --no-host-frames --no-readout --no-trajectory --steps 100 --warmup 10 2>/dev/null | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['concurrent_rooms_one_gpu']; print('v$i rep $r one room %.0f  two %.0f  four %.0f  icp %.1f' % (d['value'], c['2_rooms']['frames_per_s_in_all'], c['4_rooms']['frames_per_s_in_all'], d['stage_us']['icp']))"
  done
done
