#!/bin/bash
# usage: tools/noise_kstats.sh [N] [frames] [stream] -- kernel stats of the tracker on a stream with holes (tools/noise_run.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/noise_kstats_${3:-noise}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/noise_run.py "$@" > $OUT/log.txt 2>&1
grep "^frame\|^us/frame" $OUT/log.txt | tail -3
python3 $ROOT/tools/kstats.py $OUT 10
