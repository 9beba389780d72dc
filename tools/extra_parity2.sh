#!/bin/bash
# Second batch of library-neutral parity runs on the final build (see extra_parity.sh): whole three-turn scans of rooms 2 and 3 as
# a sensor sees them (256^3), the up and down turns of room 1 at 512^3 as a sensor sees them, 600 more fuzz seeds.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
X=$ROOT/gpurun_out/extra
mkdir -p $X; cd $ROOT
run() { out=$1; shift; timeout 1500 python3 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 > $X/$out; cat $X/$out; }
run long_parity_room2_sensor_256_whole_scan.txt tools/long_parity.py 256 721 --room 2 --sensor
run long_parity_room3_sensor_256_whole_scan.txt tools/long_parity.py 256 721 --room 3 --sensor
run long_parity_room1_sensor_down_512.txt tools/long_parity.py 512 250 --room 1 --first 470 --sensor
python3 tools/fuzz_campaign.py 6000 600 > $X/fuzz_campaign_6000.txt 2>&1; tail -1 $X/fuzz_campaign_6000.txt
