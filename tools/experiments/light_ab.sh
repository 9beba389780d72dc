#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/light_ab.sh "<extra hipcc defines>" tag [stream]  -- rebuild integrate.hip with defines, kernel medians on a holes stream
cd ${GRAFT_REPO_ROOT:-.}
touch housescan_amd/csrc/integrate.hip
make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical $1" 2>&1 | grep -E "error"
echo "== $2"
tools/noise_kstats.sh 512 40 ${3:-noise} | grep -E "integrate|us/frame"
