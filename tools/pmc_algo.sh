#!/bin/bash
# PMC passes of one MSDA kernel variant: tools/pmc_algo.sh <algo> <kernel-substring> <outdir>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A=$1; K=$2; O=$R/gpurun_out/$3
mkdir -p $O
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$O/$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/tools/run_algo.py $A 6 > $O/log.txt 2>&1 || true
done
python3 $R/tools/pmc_summary.py $O $K | tee $O/summary.txt
