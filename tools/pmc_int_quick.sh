#!/bin/bash
# usage: tools/pmc_int_quick.sh <outdir> [volume] [frames]  -- instruction-mix counters of the integrate kernels over the tracked
# stream (tools/replay_frames.py: the frames bench.py times), one rocprofv3 --pmc pass; summary printed and kept as summary.txt
OUT=$1; VOL=${2:-512}; N=${3:-26}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/$OUT/p1 -- python3 $ROOT/tools/replay_frames.py $VOL $N > $ROOT/$OUT/p1.log 2>&1
python3 $ROOT/tools/pmc_summary.py $ROOT/$OUT k_integrate k_column | tee $ROOT/$OUT/summary.txt
rm -rf $ROOT/$OUT/p1
