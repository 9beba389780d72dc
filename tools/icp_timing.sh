#!/bin/bash
source "$(dirname "$0")/restore_default.sh"
# usage: tools/icp_timing.sh ["extra -D flags"] -- per-block phase times of the 19 ICP iterations (timing build)
cd ${GRAFT_REPO_ROOT:-.}
touch housescan_amd/csrc/kernels_image.hip
make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical -DHSK_ICP_TIMING $1" 2>&1 | grep -E "error"
python tools/icp_timing.py
touch housescan_amd/csrc/kernels_image.hip; make -s -C housescan_amd/csrc 2>&1 | grep error
