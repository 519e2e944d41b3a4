#!/bin/bash
# L2 -> fabric requests of the resident-levels kernel (development library) under its switches: tiled / linear query order, batch size.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_rf
mkdir -p $O
export RDETR_LIB_PATH=$R/relation_detr_amd/librelation_detr_amd_dev.so
for V in "tiled1_B4" "tiled0_B4" "tiled1_B1" "tiled1_B8"; do
  T=${V:5:1}; B=${V:8}
  export RDETR_DEV_RES_TILED=$T RDETR_PROFILE_B=$B
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/$V -- python3 $R/tools/profile_win.py bhsd 12 auto r50 > $O/log.txt 2>&1 || { echo "FAILED $V"; tail -5 $O/log.txt; exit 1; }
  echo "== $V" | tee -a $O/pmc.txt
  python3 $R/tools/pmc_summary.py $O/$V msda_fwd_res | grep -v "^==" | tee -a $O/pmc.txt
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
