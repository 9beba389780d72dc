#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/int_timing.sh [volume] ["extra -D flags"] -- per-wave life of integrate pass B (timing build)
cd ${GRAFT_REPO_ROOT:-.}
touch housescan_amd/csrc/integrate.hip housescan_amd/csrc/raycast.hip housescan_amd/csrc/extract.hip
make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical -DHSK_INT_TIMING $2" 2>&1 | grep -E "error"
python tools/int_timing.py ${1:-512}
touch housescan_amd/csrc/integrate.hip housescan_amd/csrc/raycast.hip housescan_amd/csrc/extract.hip; make -s -C housescan_amd/csrc 2>&1 | grep error
