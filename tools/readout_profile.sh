#!/bin/bash
# usage: tools/readout_profile.sh [volume] -- kernel stats of the read-out kernels (k_extract*, k_scan_rows_*, k_summaries,
# k_vol_convert): bench.py with only the read-out block after a short timed region, under rocprofv3 --kernel-trace --stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-512}
OUT=$ROOT/gpurun_out/readout_$V
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --volume $V --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-1024 --no-rooms --no-trajectory --no-noise --no-host-frames > $OUT/log.txt 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
grep -o '{"metric.*' $OUT/log.txt | python3 -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read())['readout_ms'], indent=1))" > $OUT/readout_ms.json
python3 $ROOT/tools/kstats.py $OUT/trace 40 | grep -i "extract\|scan_rows\|summaries\|vol_convert\|rebuild"
cat $OUT/readout_ms.json
rm -rf $OUT/trace
