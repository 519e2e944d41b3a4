#!/bin/bash
# Same-box A/B of the direct MSDA kernel with its query-side streams / output rows plain vs non-temporal (make -C relation_detr_amd/csrc nt):
# durations (kernel trace) and HBM-side traffic (FETCH_SIZE, WRITE_SIZE) at the FocalNet (B = 2) and R50 (B = 4) encoder shapes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_nt
mkdir -p $O
for CFG in focalnet r50; do
  for LIB in plain nt; do
    if [ $LIB = nt ]; then export RDETR_LIB_PATH=$R/relation_detr_amd/librelation_detr_amd_nt.so; else unset RDETR_LIB_PATH; fi
    for C in "--kernel-trace --stats" "--pmc FETCH_SIZE --kernel-trace" "--pmc WRITE_SIZE --kernel-trace" "--pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace"; do
      T=$(echo $C | tr -d ' -' | cut -c1-24)
      timeout -k 10 200 rocprofv3 $C --output-format csv -d $O/${CFG}_${LIB}_$T -- python3 $R/tools/profile_win.py bhsd 12 direct $CFG > $O/log.txt 2>&1 || echo "FAILED $CFG $LIB $C"
    done
    echo "== $CFG $LIB"
    python3 $R/tools/pmc_summary.py $O/${CFG}_${LIB}_pmcFETCH_SIZEkerneltrace msda_fwd_qrun | grep -v "^=="
    python3 $R/tools/pmc_summary.py $O/${CFG}_${LIB}_pmcWRITE_SIZEkerneltrace msda_fwd_qrun | grep "WRITE"
    python3 $R/tools/pmc_summary.py $O/${CFG}_${LIB}_pmcTCC_HIT_sumTCC_MISS_s msda_fwd_qrun | grep "TCC"
    grep -h "msda_fwd_qrun" $O/${CFG}_${LIB}_kerneltracestats/*/*kernel_stats.csv | cut -d, -f2-6
  done
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
