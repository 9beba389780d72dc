#!/bin/bash
# usage: tools/trace_run.sh <bench args...> -- kernel trace of a short bench run, then the idle-gap table
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python3 $ROOT/tools/gaps.py $OUT 0.4
