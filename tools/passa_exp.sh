#!/bin/bash
# pass A (k_integrate) time split: debug builds, only the kernel time is read (rocprofv3 stats); results are wrong by construction
cd ${GRAFT_REPO_ROOT:-.}
for v in "" "-DHSK_EXPA_NO_L2" "-DHSK_EXPA_NO_FREE" "-DHSK_EXPA_NO_QUEUE" "-DHSK_EXPA_NO_L2 -DHSK_EXPA_NO_FREE -DHSK_EXPA_NO_QUEUE"; do
  touch housescan_amd/csrc/kernels_volume.hip
  make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function $v" 2>&1 | grep -E "error"
  echo "== [$v]"
  tools/kstats_run.sh --steps 60 --warmup 10 2>&1 | grep -E "k_integrate<false>|k_integrate_detail<false>"
done
touch housescan_amd/csrc/kernels_volume.hip; make -s -C housescan_amd/csrc 2>&1 | grep error
