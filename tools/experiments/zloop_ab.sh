#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
cd ${GRAFT_REPO_ROOT:-.}
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
cp housescan_amd/libhskinfu.so /tmp/lib_default.so
make -s -j8 -C housescan_amd/csrc FLAGS="$BASE -DHSK_PA_ZLOOP -DINTEGRATE_WPE=7 -DINTEGRATE_WPE_LONG=5" 2>&1 | grep -E "error"
cp housescan_amd/libhskinfu.so /tmp/lib_zloop.so
HSK_PA_ZSPLIT=4 python tools/quick_parity.py 128 256 2>&1 | tail -1
HSK_PA_ZSPLIT=16 python tools/quick_parity.py 512 2>&1 | tail -1
run() { python bench.py --allow-exp --quick --steps 60 --warmup 10 --volume $1 2>/dev/null | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_us']; print('$2 vol $1 fps %.0f icp %.1f integrate %.1f raycast %.1f frac %.3f' % (d['value'], s['icp'], s['integrate'], s['raycast'], d['roofline']['frac']))"; }
for rep in 1 2; do
  cp /tmp/lib_default.so housescan_amd/libhskinfu.so; run 512 default
  cp /tmp/lib_zloop.so housescan_amd/libhskinfu.so
  for z in 64 32 16 8; do HSK_PA_ZSPLIT=$z run 512 zsplit$z; done
done
for rep in 1 2; do
  cp /tmp/lib_default.so housescan_amd/libhskinfu.so; run 1024 default
  cp /tmp/lib_zloop.so housescan_amd/libhskinfu.so
  for z in 64 32 16 8; do HSK_PA_ZSPLIT=$z run 1024 zsplit$z; done
done
