#!/bin/bash
# PMC counters of the integrate kernels on the integrate-only bench (tools/int_bench.py), one rocprofv3 pass per group,
# each under a timeout (the SPI_* counters abort rocprofv3 and hang: never add them).
# usage: tools/pmc_int.sh <outdir under gpurun_out> [volume] [groups: all|mem|tlb]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; VOL=${2:-512}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
# (at most a few counters of one hardware block per pass: "Request exceeds the capabilities of the hardware" aborts rocprofv3)
G4="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
G5="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_TC_BUSY GRBM_EA_BUSY"
G6="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_TAG_STALL_sum"
G7="TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TD_TD_BUSY_sum SQ_VMEM_TA_ADDR_FIFO_FULL"
G8="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY"
G9="TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"
i=0
for grp in "$G1" "$G2" "$G4" "$G5" "$G6" "$G7" "$G8" "$G9"; do
  i=$((i+1))
  if [ "$3" = "mem" ] && { [ $i -le 2 ] || [ $i -ge 7 ]; }; then continue; fi
  if [ "$3" = "tlb" ] && [ $i -le 6 ]; then continue; fi
  timeout 60 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/int_bench.py $VOL 0 16 > $OUT/p$i.log 2>&1 || echo "group $i failed or timed out"
done
python3 $ROOT/tools/pmc_summary.py $OUT k_integrate k_column > $OUT/summary.txt 2>&1
rm -rf $OUT/p[0-9]*/
grep -v "^traffic" $OUT/summary.txt
