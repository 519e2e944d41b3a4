#!/bin/bash
# Same-box A/B of the resident-levels kernel (development library): planes of an XCD in flight at a time -- times and L2 -> fabric requests.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_rt
mkdir -p $O
export RDETR_LIB_PATH=$R/relation_detr_amd/librelation_detr_amd_dev.so
timeout -k 10 300 python3 $R/tools/res_check.py r50:4 r50:2 2>&1 | grep -v amdgpu | cut -c1-100 | tee $O/times.txt
for F in 0 1; do
  for MT in 0 2 1; do
    export RDETR_DEV_RES_MAX_TEAMS=$MT RDETR_PROFILE_FUSED=$F
    timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/f${F}_mt${MT} -- python3 $R/tools/profile_win.py bhsd 12 auto r50 > $O/log.txt 2>&1 || { echo "FAILED $F $MT"; tail -5 $O/log.txt; exit 1; }
    echo "== fused=$F planes in flight per XCD=$MT (0 = all)" | tee -a $O/pmc.txt
    python3 $R/tools/pmc_summary.py $O/f${F}_mt${MT} msda_fwd_res | grep -v "^==" | tee -a $O/pmc.txt
  done
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
