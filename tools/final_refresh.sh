#!/bin/bash
# Round-end evidence run on one MI355X box (one gpurun call): GPU tests, smoke, default bench (bf16, with CPU baseline and the
# fp32 / 300-query side runs), fp32 bench, one-stream and one-rank-RCCL variants, the rocprofv3 kernel summary of the bench
# command (graph replay, as benched) and a second one with one stream, eager, no GEMM tuning (clean per-kernel durations), and
# the PMC passes of the benched gather kernel.  Output: gpurun_out/final/ (copy what is judged into profiles/rNN/).
# GEMM tuning (TunableOp) is opt-in since round 3: every run here uses the library's own heuristics.
set -e
O=gpurun_out/final
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > $O/bench_default_bf16.json.log 2>$O/bench_default_bf16.err
tail -1 $O/bench_default_bf16.json.log | cut -c1-300
python3 bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json.log 2>/dev/null
tail -1 $O/bench_fp32.json.log | cut -c1-200
RDETR_BENCH_STREAMS=1 python3 bench.py --no-cpu-baseline --no-extras > $O/bench_bf16_one_stream.json.log 2>/dev/null
RDETR_BENCH_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_bf16_one_rank_rccl.json.log 2>/dev/null
python3 bench.py --config focalnet --steps 10 --warmup 3 > $O/bench_focalnet_bf16.json.log 2>$O/bench_focalnet_bf16.err
tail -1 $O/bench_focalnet_bf16.json.log | cut -c1-200
tail -1 $O/bench_bf16_one_rank_rccl.json.log | cut -c1-200
tail -1 $O/bench_bf16_one_stream.json.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_as_benched -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_under_rocprofv3_bf16.json.log 2>$O/bench_under_rocprofv3_bf16.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
export RDETR_BENCH_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16_eager_one_stream -- python3 bench.py --steps 16 --warmup 4 --no-graph --no-cpu-baseline --no-extras > $O/bench_profiled_bf16_eager.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
unset RDETR_BENCH_STREAMS
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- python3 tools/profile_win.py bhsd 6 auto > $O/pmc_$tag.log 2>&1
done
find $O -name "*agent_info.csv" -delete
du -sh $O
