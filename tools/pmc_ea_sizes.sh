#!/bin/bash
# Bytes that leave the L2s, by request size (TCC_EA0_RDREQ_{32B,64B,128B}): calibrates FETCH_SIZE for the direct MSDA kernel's mix of
# streamed 16-byte loads and gathered 64-byte rows (MI355X_MICROARCH.md: FETCH_SIZE halves wide streaming reads, other widths uncalibrated).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ea
mkdir -p $O
for CFG in focalnet r50; do
  for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr -d ' ' | cut -c1-30)
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${CFG}_$T -- python3 $R/tools/profile_win.py bhsd 10 direct $CFG > $O/log.txt 2>&1 || echo "FAILED $CFG $C"
    echo "== $CFG"; python3 $R/tools/pmc_summary.py $O/${CFG}_$T msda_fwd_qrun | grep -v "^==\|_dur"
  done
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
