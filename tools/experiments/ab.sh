#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/ab.sh "<extra hipcc defines>" tag   -- rebuild the kernels with defines and run a short bench (GPU box)
cd ${GRAFT_REPO_ROOT:-.}
touch housescan_amd/csrc/integrate.hip housescan_amd/csrc/raycast.hip housescan_amd/csrc/extract.hip housescan_amd/csrc/kernels_image.hip
make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical $1" 2>&1 | grep -E "error"
python bench.py --allow-exp --steps 60 --warmup 10 --quick 2>&1 | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', 'fps', d['value'], 'stage_us', {k:v for k,v in d['stage_us'].items() if k!='note'}, 'roofline GB/s', d['roofline']['achieved'])"
