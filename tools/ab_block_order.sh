#!/bin/bash
# Same-box A/B of the direct MSDA kernel's block order at the encoder shape: query order (identity) vs band-interleaved levels.
# Development library (make -C relation_detr_amd/csrc dev).  Durations from the kernel trace, L2 -> fabric read requests from PMC.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_order
mkdir -p $O
export RDETR_LIB_PATH=$R/relation_detr_amd/librelation_detr_amd_dev.so
for CFG in focalnet r50; do
  for ORD in identity interleaved; do
    if [ $ORD = identity ]; then export RDETR_DEV_IDENTITY_ORDER=1; else unset RDETR_DEV_IDENTITY_ORDER; fi
    for C in "--kernel-trace --stats" "--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace"; do
      T=$(echo $C | tr -d ' -' | cut -c1-16)
      timeout -k 10 200 rocprofv3 $C --output-format csv -d $O/${CFG}_${ORD}_$T -- python3 $R/tools/profile_win.py bhsd 40 direct $CFG > $O/log.txt 2>&1 || echo "FAILED $CFG $ORD $C"
    done
    echo "== $CFG $ORD"
    python3 $R/tools/pmc_summary.py $O/${CFG}_${ORD}_pmcTCC_EA0_RDREQ msda_fwd_qrun | grep -v "^=="
    grep -h "msda_fwd_qrun" $O/${CFG}_${ORD}_kerneltracestat/*/*kernel_stats.csv | sed 's/.*)",//' 
  done
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
