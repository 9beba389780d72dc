#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/ab_repeat.sh REPS "<defines A>" "<defines B>" ...  -- each variant built once, benched REPS times in turn
# (no profiler attached: the stage times come from the library's own HIP events), variants interleaved so that box
# drift hits them alike
cd ${GRAFT_REPO_ROOT:-.}
reps=$1; shift
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
i=0
for defs in "$@"; do
  make -s -j8 -C housescan_amd/csrc FLAGS="$BASE $defs" 2>&1 | grep -E "error"
  cp housescan_amd/libhskinfu.so /tmp/libhsk_v$i.so
  echo "v$i = [$defs]"
  # PARITY="128 256": integrate parity of every variant against the oracle before it is timed (tools/quick_parity.py)
  if [ -n "$PARITY" ]; then python tools/quick_parity.py $PARITY 2>&1 | tail -1; fi
  i=$((i+1))
done
n=$i
for r in $(seq 1 $reps); do
  for i in $(seq 0 $((n-1))); do
    cp /tmp/libhsk_v$i.so housescan_amd/libhskinfu.so
    python bench.py --allow-exp --quick --steps ${STEPS:-60} --warmup 10 --volume ${VOL:-512} 2>/dev/null | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_us']; print('v$i rep $r fps %.0f icp %.1f integrate %.1f raycast %.1f frac %.3f' % (d['value'], s['icp'], s['integrate'], s['raycast'], d['roofline']['frac']))"
  done
done
