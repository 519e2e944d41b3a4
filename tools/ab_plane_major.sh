#!/bin/bash
# Same-box A/B of the resident-levels kernel's plane -> XCD map (development library): times (tools/res_check.py) and L2 -> fabric
# requests (PMC) for the operator and the fused form.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_pm
mkdir -p $O
export RDETR_LIB_PATH=$R/relation_detr_amd/librelation_detr_amd_dev.so
timeout -k 10 300 python3 $R/tools/res_check.py r50:4 r50:2 2>&1 | grep -v amdgpu | cut -c1-100 | tee $O/times.txt
for F in 0 1; do
  for PM in 0 1; do
    export RDETR_DEV_RES_PLANE_MAJOR=$PM RDETR_PROFILE_FUSED=$F
    timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/f${F}_pm${PM} -- python3 $R/tools/profile_win.py bhsd 12 auto r50 > $O/log.txt 2>&1 || { echo "FAILED $F $PM"; tail -5 $O/log.txt; exit 1; }
    echo "== fused=$F plane-major=$PM" | tee -a $O/pmc.txt
    python3 $R/tools/pmc_summary.py $O/f${F}_pm${PM} msda_fwd_res | grep -v "^==" | tee -a $O/pmc.txt
  done
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
