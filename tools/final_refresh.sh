#!/bin/bash
# Round-end evidence run on one MI355X box: GPU tests, smoke, default bench (bf16, with CPU baseline), fp32 bench, and the
# rocprofv3 kernel summary of the bench command (one stream, eager, no GEMM tuning: clean per-kernel durations).
set -e
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1 || { tail -30 gpurun_out/final/gpu_tests.log; exit 1; }
tail -2 gpurun_out/final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > gpurun_out/final/bench_default_bf16.json.log 2>gpurun_out/final/bench_default_bf16.err
tail -1 gpurun_out/final/bench_default_bf16.json.log | cut -c1-260
python3 bench.py --dtype fp32 --no-cpu-baseline > gpurun_out/final/bench_fp32.json.log 2>/dev/null
tail -1 gpurun_out/final/bench_fp32.json.log | cut -c1-260
RDETR_BENCH_STREAMS=1 python3 bench.py --no-cpu-baseline > gpurun_out/final/bench_bf16_one_stream.json.log 2>/dev/null
RDETR_BENCH_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/bench_bf16_one_rank_rccl.json.log 2>/dev/null
tail -1 gpurun_out/final/bench_bf16_one_rank_rccl.json.log | cut -c1-200
tail -1 gpurun_out/final/bench_bf16_one_stream.json.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export RDETR_BENCH_TUNABLEOP=0 RDETR_BENCH_STREAMS=1 RDETR_BENCH_ALT300=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_bf16 -- python3 bench.py --steps 16 --warmup 4 --no-graph --no-cpu-baseline > gpurun_out/final/bench_profiled_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_fp32 -- python3 bench.py --steps 8 --warmup 2 --dtype fp32 --no-graph --no-cpu-baseline > gpurun_out/final/bench_profiled_fp32.log 2>&1
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*.csv" | head
