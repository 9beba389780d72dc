#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/pb_timing.sh [volume] ["extra -D flags"] -- per-wave phases of integrate pass B (timing build)
cd ${GRAFT_REPO_ROOT:-.}
make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical -DHSK_PB_TIMING $2" 2>&1 | grep -E "error"
python tools/pb_timing.py ${1:-512}
