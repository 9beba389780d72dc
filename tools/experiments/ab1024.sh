#!/bin/bash
source "$(dirname "$0")/../restore_default.sh"
# usage: tools/experiments/ab1024.sh "<defines>" ... -- rebuild with each define set, bench 512^3 and 1024^3 (GPU box)
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  touch housescan_amd/csrc/integrate.hip housescan_amd/csrc/raycast.hip housescan_amd/csrc/extract.hip housescan_amd/csrc/kernels_image.hip
  make -s -C housescan_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-function -Wno-bitwise-instead-of-logical $v" 2>&1 | grep -E "error"
  for n in 512 1024; do
    python bench.py --allow-exp --steps 60 --warmup 10 --no-cpu-baseline --volume $n 2>&1 | grep -o '{"metric.*' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v]', $n, 'fps', d['value'], 'integrate', d['stage_us']['integrate'], 'raycast', d['stage_us']['raycast'])"
  done
done
touch housescan_amd/csrc/integrate.hip housescan_amd/csrc/raycast.hip housescan_amd/csrc/extract.hip housescan_amd/csrc/kernels_image.hip; make -s -C housescan_amd/csrc 2>&1 | grep error
