#!/bin/bash
# Same-box A/B of the direct MSDA kernel's head-group size: times (tools/ab_head_group.py) and L2 -> fabric read requests (PMC).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_hg
mkdir -p $O
export RDETR_LIB_PATH=$R/relation_detr_amd/librelation_detr_amd_dev.so
timeout -k 10 400 python3 $R/tools/ab_head_group.py 2>&1 | tee $O/times.txt || exit 1
for CFG in r50 focalnet; do
  for HG in 0 1 2 3; do
    export RDETR_DEV_HEAD_GROUP_LOG2=$HG
    timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${CFG}_hg${HG} -- python3 $R/tools/profile_win.py bhsd 12 direct $CFG > $O/log.txt 2>&1 || { echo "FAILED $CFG $HG"; tail -5 $O/log.txt; exit 1; }
    echo "== $CFG G=$((1 << HG))" | tee -a $O/pmc.txt
    python3 $R/tools/pmc_summary.py $O/${CFG}_hg${HG} msda_fwd_qrun | grep -v "^==" | tee -a $O/pmc.txt
  done
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
