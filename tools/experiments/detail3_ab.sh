#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/detail3_ab.sh -- pass B's third form: grid size (DETAIL3_GX x 256 blocks of 4 waves) and the register budget
# (DETAIL3_WPE waves per SIMD): quick parity, then pass B's median on the scripted stream, the noise run and 1024^3
cd ${GRAFT_REPO_ROOT:-.}
for fl in ${SWEEP:-"" "-DDETAIL3_GX=4" "-DDETAIL3_GX=6" "-DDETAIL3_GX=12" "-DDETAIL3_GX=16" "-DDETAIL3_GX=24" "-DDETAIL3_GX=32" "-DDETAIL3_GX=48"}; do
  touch housescan_amd/csrc/integrate.hip
  make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical $fl" 2>&1 | grep -E "error"
  echo "== [$fl]"; python tools/quick_parity.py 128 | tail -1
  for job in "512 40 scripted" "512 40 noise" "1024 24 scripted"; do set -- $job; echo "-- $3 $1: $(tools/noise_kstats.sh $1 $2 $3 | grep -E 'k_integrate_detail3<false' | sed 's/.*median= *\([0-9.]*\).*/pass B median \1 us/')"; done
done
