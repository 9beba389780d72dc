#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/ab_files.sh REPS lib_a.so lib_b.so ...  -- like ab_repeat.sh for libraries built beforehand (e.g. one from
# another commit), benched REPS times in turn
cd ${GRAFT_REPO_ROOT:-.}
reps=$1; shift
for r in $(seq 1 $reps); do
  i=0
  for f in "$@"; do
    cp "$f" housescan_amd/libhskinfu.so
    python bench.py --allow-exp --quick --steps ${STEPS:-60} --warmup 10 --volume ${VOL:-512} 2>/dev/null | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_us']; print('v$i rep $r fps %.0f icp %.1f integrate %.1f raycast %.1f frac %.3f' % (d['value'], s['icp'], s['integrate'], s['raycast'], d['roofline']['frac']))"
    i=$((i+1))
  done
done
