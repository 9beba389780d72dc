#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
cd ${GRAFT_REPO_ROOT:-.}
i=0
for fl in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause"; do
  make -s -j8 -C housescan_amd/csrc IMAGE_FLAGS="$fl" 2>&1 | grep -E "error"
  cp housescan_amd/libhskinfu.so /tmp/libimg_v$i.so; echo "v$i = [$fl]"; python tools/quick_parity.py 128 2>&1 | tail -1
  i=$((i+1))
done
for r in 1 2 3 4; do for i in 0 1 2; do
  cp /tmp/libimg_v$i.so housescan_amd/libhskinfu.so
  python bench.py --allow-exp --quick --steps 60 --warmup 10 2>/dev/null | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_us']; print('v$i rep $r fps %.0f icp %.1f pre %.1f' % (d['value'], s['icp'], s['preprocess']))"
done; done
