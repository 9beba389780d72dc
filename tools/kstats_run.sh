#!/bin/bash
# usage: tools/kstats_run.sh <bench args...>  -- rocprofv3 kernel stats of a short bench run, top kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --quick "$@" > $OUT/log.txt 2>&1
python3 $ROOT/tools/kstats.py $OUT 12
