#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/zwg_ab.sh "1 2 4"  -- pass A with PA_ZWG wave-chunk quadruples per workgroup: quick parity, then kernel medians on the
# scripted stream, the room scan and (scripted) 1024^3
cd ${GRAFT_REPO_ROOT:-.}
for Z in $1; do
  touch housescan_amd/csrc/integrate.hip
  make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical -DPA_ZWG=$Z" 2>&1 | grep -E "error"
  echo "== PA_ZWG=$Z"
  python tools/quick_parity.py 64 128 256 | tail -1
  for st in scripted room0 noise; do echo "-- $st 512"; tools/noise_kstats.sh 512 40 $st | grep -E "k_integrate<|us/frame"; done
  echo "-- scripted 1024"; tools/noise_kstats.sh 1024 24 scripted | grep -E "k_integrate<|us/frame"
done
