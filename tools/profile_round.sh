#!/bin/bash
# Collects the round's judged profile artefacts on the GPU box:
#   gpurun_out/round/kernel_stats.csv     rocprofv3 --kernel-trace --stats of `python bench.py` (default flags)
#   gpurun_out/round/bench.json           the JSON line of that run
#   gpurun_out/round/pmc/...              FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs) + integrate_traffic.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/round
rm -rf $OUT/trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --quick > $OUT/bench_profiled.log 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
grep -o '{"metric.*' $OUT/bench_profiled.log > $OUT/bench_profiled.json
cd $ROOT && tools/pmc.sh gpurun_out/round/pmc > $OUT/pmc_summary.txt 2>&1
cp $OUT/pmc/integrate_traffic.json $OUT/ 2>/dev/null
# the same at 1024^3 (the HBM measurement): kernel averages of a 40-frame run
rm -rf $OUT/trace1024
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1024 -- python3 $ROOT/bench.py --quick --volume 1024 --steps 40 --warmup 5 > $OUT/bench_profiled_1024.log 2>&1)
cp $OUT/trace1024/*/*_kernel_stats.csv $OUT/kernel_stats_1024.csv
grep -o '{"metric.*' $OUT/bench_profiled_1024.log > $OUT/bench_profiled_1024.json
rm -rf $OUT/trace1024
# long-horizon parity of THIS build against the oracle's tracker: every pose and the final TSDF, bit for bit
for job in "256 300" "512 300" "1024 100"; do
  set -- $job
  (cd $ROOT && timeout 1200 python3 tools/long_parity.py $1 $2 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_$1.txt)
  cat $OUT/long_parity_$1.txt
done
# ... on the noise run (sigma = 1.2 mm z^2, 2 % dropout), where the hole-aware paths of integrate do their work
(cd $ROOT && timeout 1200 python3 tools/long_parity.py 512 300 --noise 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_noise_512.txt)
cat $OUT/long_parity_noise_512.txt
# (round 6) ... on the ROOM SCAN -- camera inside the volume: the level turn, the up turn and the start of the down turn of room 0 --
# and on the stream with holes as a sensor makes them
(cd $ROOT && timeout 1500 python3 tools/long_parity.py 512 520 --room 0 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_room_512.txt)
cat $OUT/long_parity_room_512.txt
(cd $ROOT && timeout 1500 python3 tools/long_parity.py 256 721 --room 1 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_room_256_whole_scan.txt)
cat $OUT/long_parity_room_256_whole_scan.txt
(cd $ROOT && timeout 1200 python3 tools/long_parity.py 512 300 --holes 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_holes_512.txt)
cat $OUT/long_parity_holes_512.txt
# ... the other two rooms (level turn and into the up turn), and a room as a SENSOR sees it (holes inside the volume's own frustum)
for v in 2 3; do
  (cd $ROOT && timeout 1200 python3 tools/long_parity.py 512 300 --room $v 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_room${v}_512.txt)
  cat $OUT/long_parity_room${v}_512.txt
done
(cd $ROOT && timeout 1200 python3 tools/long_parity.py 512 300 --room 0 --sensor 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_room_sensor_512.txt)
cat $OUT/long_parity_room_sensor_512.txt
# ... kernel medians on the four streams (tools/noise_kstats.sh) and the rooms from C threads with their kernel overlap
for st in scripted holes noise room0; do
  (cd $ROOT && timeout 600 tools/noise_kstats.sh 512 40 $st > $OUT/kernel_medians_512_$st.txt 2>&1)
done
(cd $ROOT && ROOMS="1 2 4 8" TRACE=4 timeout 900 tools/rooms_native.sh 512 240 room 1 host 0 > $OUT/rooms_native_room.txt 2>&1)
(cd $ROOT && ROOMS="1 2 4" timeout 900 tools/rooms_native.sh 512 240 open 1 host 0 > $OUT/rooms_native_open.txt 2>&1)
(cd $ROOT && ROOMS="1 4" timeout 900 tools/rooms_native.sh 512 240 room 1 host 2 > $OUT/rooms_native_room_graph.txt 2>&1)
[ -x $ROOT/tools/probes/launch_gap_probe ] || (cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $ROOT/tools/probes/launch_gap_probe.hip -o $ROOT/tools/probes/launch_gap_probe 2>/dev/null)
(cd $ROOT && timeout 120 tools/probes/launch_gap_probe > $OUT/launch_gap_probe.txt 2>&1)
# the read-out kernels (not in the bench's timed region): kernel stats + host times at both sizes
for v in 512 1024; do
  (cd $ROOT && timeout 900 tools/readout_profile.sh $v > $OUT/readout_$v.txt 2>&1)
  cp $ROOT/gpurun_out/readout_$v/kernel_stats.csv $OUT/readout_kernel_stats_$v.csv 2>/dev/null
done
# ... and of the z-slab group on this one device: 2 and 8 slabs with the direct exchange, 4 with the staged composites,
# 2 with the row-sharded ICP (N slabs = one volume, bit for bit, at length)
for job in "512 300 2 direct" "512 300 8 direct" "512 100 4 composite" "512 100 2 icp_allreduce" "1024 60 4 direct" "1024 60 8 direct"; do
  set -- $job
  (cd $ROOT && timeout 1200 python3 tools/long_parity_slabs.py $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_slabs_$1_$3_$4.txt)
  cat $OUT/long_parity_slabs_$1_$3_$4.txt
done
# ... and with every configuration field off its default (non-cubic, non-power-of-two volume, fx != fy, other gates and iteration counts)
(cd $ROOT && timeout 1200 python3 tools/long_parity_nondefault.py 300 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_nondefault.txt)
cat $OUT/long_parity_nondefault.txt
# ... the same configuration on frames as a sensor returns them: the light class of pass A / pass B away from every default
(cd $ROOT && timeout 1200 python3 tools/long_parity_nondefault.py 300 --holes 2>&1 | grep -v amdgpu.ids | tail -4 > $OUT/long_parity_nondefault_holes.txt)
cat $OUT/long_parity_nondefault_holes.txt
# the raw per-dispatch CSVs exceed what gpurun copies back (64 MiB): only the summaries travel
rm -rf $OUT/trace $OUT/pmc/p[0-9]*
tail -5 $OUT/pmc_summary.txt
