set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02/evidence
mkdir -p $O
cd $R
# 1. kernel trace + stats of the bench command (side measurements off: the fp32 child process is not started under the profiler)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.log 2> $O/bench_under_rocprof.err
find $O/bench_stats -name '*kernel_trace.csv' -delete; find $O/bench_stats -name '*agent_info.csv' -delete   # keep the stats summary only (the trace is > 64 MiB)
# 2. PMC passes of the benched gather kernel (direct, head-major), one counter group per pass
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- python3 tools/profile_win.py bhsd 6 direct > $O/pmc_$tag.log 2>&1
done
find $O -name '*agent_info.csv' -delete; du -sh $O
# 3. plain bench (default command)
timeout -k 10 400 python3 bench.py > $O/bench_default.log 2> $O/bench_default.err
tail -c 1500 $O/bench_default.log
