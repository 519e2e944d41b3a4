#!/bin/bash
# PMC passes (one counter group per pass) of the 5-level gather kernel at the FocalNet encoder shape (B = 2) -> gpurun_out/final/pmc_focalnet/
O=gpurun_out/final/pmc_focalnet
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- python3 tools/profile_win.py bhsd 6 auto focalnet > $O/pmc_$tag.log 2>&1 || exit 1
done
find $O -name "*agent_info.csv" -delete
du -sh $O
