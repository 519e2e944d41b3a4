#!/bin/bash
# extra PMC passes of one MSDA kernel variant (texture path / L1 side): tools/pmc_algo2.sh <algo> <kernel-substring> <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A=$1; K=$2; O=$R/gpurun_out/$3
mkdir -p $O
for C in "TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_BUSY_max" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  D=$O/$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/tools/run_algo.py $A 6 > $O/log.txt 2>&1 || echo "pass failed: $C"
done
python3 $R/tools/pmc_summary.py $O $K | tee $O/summary.txt
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
