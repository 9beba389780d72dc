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
# the raw per-dispatch CSVs exceed what gpurun copies back (64 MiB): only the summaries travel
rm -rf $OUT/trace $OUT/pmc/p[0-9]*
tail -5 $OUT/pmc_summary.txt
