#!/bin/bash
# Library-neutral extra evidence on the round's final build (run after tools/final_round.sh; gpurun_out/extra -> profiles/rNN by hand):
# the light class inside slab contexts at length, the WHOLE three-turn scan of a room at 512^3, holes and noise at 1024^3,
# and another range of fuzz seeds.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
X=$ROOT/gpurun_out/extra
mkdir -p $X; cd $ROOT
run() { out=$1; shift; timeout 1500 python3 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 > $X/$out; cat $X/$out; }
run long_parity_slabs_512_4_direct_holes.txt tools/long_parity_slabs.py 512 300 4 direct --holes
run long_parity_slabs_512_8_direct_noise.txt tools/long_parity_slabs.py 512 200 8 direct --noise
run long_parity_slabs_512_3_composite_noise.txt tools/long_parity_slabs.py 512 100 3 composite --noise
run long_parity_room_512_whole_scan.txt tools/long_parity.py 512 721 --room 0
run long_parity_holes_1024.txt tools/long_parity.py 1024 60 --holes
run long_parity_noise_1024.txt tools/long_parity.py 1024 60 --noise
python3 tools/fuzz_campaign.py 5000 400 > $X/fuzz_campaign_5000.txt 2>&1; tail -1 $X/fuzz_campaign_5000.txt
