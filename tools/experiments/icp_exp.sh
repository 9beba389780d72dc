#!/bin/bash
# ICP tuning A/B (debug builds; only the timing is read)
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  tools/experiments/ab.sh "$v" "[$v]"
done
touch housescan_amd/csrc/kernels_image.hip; make -s -C housescan_amd/csrc 2>&1 | grep error
