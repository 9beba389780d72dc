#!/bin/bash
# usage: tools_ab.sh "<extra hipcc defines>" tag   -- rebuild kernels_volume with defines and run a short bench
cd $GRAFT_REPO_ROOT
touch housescan_amd/csrc/kernels_volume.hip housescan_amd/csrc/kernels_image.hip
make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function $1" 2>&1 | grep -E "error" 
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', 'fps', d['value'], 'stage_us', d['stage_us'], 'roof', d['roofline']['achieved'], 'GB/s', d['roofline']['avg_launch_us'],'us')"
