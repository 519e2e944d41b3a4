#!/bin/bash
# same-box A/B of one environment switch of the harness:  tools/ab_env.sh NAME [values...]   (default values: 0 1 0 1)
name=$1; shift
for v in ${@:-0 1 0 1}; do
  env $name=$v RDETR_BENCH_ALT300=0 python3 bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name=$v', round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms')" || echo "$name=$v failed"
done
