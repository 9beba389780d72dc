#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/icp_shape_ab.sh -- block shapes of k_icp_iter (threads per block, pixels per lane at the fine / middle level) through
# IMAGE_FLAGS: the tracker's smoke parity, then the ICP stage's microseconds over 3 repetitions of a 60-frame bench each
cd ${GRAFT_REPO_ROOT:-.}
i=0
for fl in "" "-DICP_BLOCK=512 -DICP_PX_FINE=3 -DICP_PX_MID=1" "-DICP_BLOCK=512 -DICP_PX_FINE=2 -DICP_PX_MID=1" "-DICP_BLOCK=1024 -DICP_PX_FINE=2 -DICP_PX_MID=1" "-DICP_BLOCK=256 -DICP_PX_FINE=5 -DICP_PX_MID=1" "-DICP_BLOCK=128 -DICP_PX_FINE=10 -DICP_PX_MID=4"; do
  make -s -j8 -C housescan_amd/csrc IMAGE_FLAGS="$fl" 2>&1 | grep -E "error"
  cp housescan_amd/libhskinfu.so /tmp/libicp_v$i.so; echo "v$i = [$fl]"; python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
  i=$((i+1))
done
for r in 1 2 3; do for v in 0 1 2 3 4 5; do
  cp /tmp/libicp_v$v.so housescan_amd/libhskinfu.so
  python bench.py --allow-exp --quick --steps 60 --warmup 10 2>/dev/null | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_us']; print('v$v rep $r fps %.0f icp %.1f per-iter %s' % (d['value'], s['icp'], {k:v for k,v in d['icp_us_per_iter'].items() if k!='note'}))"
done; done
