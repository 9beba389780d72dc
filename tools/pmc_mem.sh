#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TCC|TCA|TA|TD)_[A-Za-z0-9_]+" | sort -u | grep -iE "UTCL|LATENCY|LEVEL|STALL|BUSY|TAGCONFLICT|EA0_RDREQ|EA0_WRREQ|HIT|MISS|PENDING" | tr '\n' ' ' | head -c 3000; echo
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $ROOT/gpurun_out/pmcm/p$i -- python3 $ROOT/bench.py --no-cpu-baseline --steps 12 --warmup 3 > $ROOT/gpurun_out/pmcm_p$i.log 2>&1
  tail -2 $ROOT/gpurun_out/pmcm_p$i.log | cut -c1-200
done
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmcm "k_integrate<false"
