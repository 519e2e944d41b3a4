set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_merged.log 2>&1 || { tail -30 gpurun_out/gpu_tests_merged.log; exit 1; }
tail -2 gpurun_out/gpu_tests_merged.log
for i in 1 2; do
RDETR_MERGED_PROJ=0 python bench.py --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split ', d['value'], d['ms_per_step'])"
RDETR_MERGED_PROJ=1 python bench.py --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merged', d['value'], d['ms_per_step'])"
done
