#!/bin/bash
# rocprofv3 kernel summary of the FocalNet (5-level) bench configuration, one stream, eager: clean per-kernel durations.
set -e
O=gpurun_out/r03/focalnet_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RDETR_BENCH_STREAMS=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --config focalnet --steps 6 --warmup 2 --no-graph --no-cpu-baseline --no-extras > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
tail -1 $O/bench.log | cut -c1-300
