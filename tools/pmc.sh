#!/bin/bash
# Collect rocprofv3 PMC counters for a short bench run, one pass per counter group (gpurun refuses --pmc with
# trace domains; FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc.sh <outdir> [bench args...]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $ROOT/$OUT/p[0-9]*; mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $ROOT/$OUT/p$i -- python3 $ROOT/bench.py --quick "$@" > $ROOT/$OUT/p$i.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $ROOT/$OUT
