#!/bin/bash
# usage: tools/pmc_quick.sh <outdir> <kernel filter> ; one pass of SQ counters
OUT=$1; FIL=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/$OUT/p1 -- python3 $ROOT/bench.py --no-cpu-baseline --steps 12 --warmup 3 > $ROOT/$OUT/p1.log 2>&1
python3 $ROOT/tools/pmc_summary.py $ROOT/$OUT "$FIL"
