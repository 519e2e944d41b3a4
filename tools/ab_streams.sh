#!/bin/bash
# same-box A/B of RDETR_BENCH_STREAMS (image groups on parallel HIP streams inside the graph)
for n in 1 2 4 1 2; do
  RDETR_BENCH_STREAMS=$n RDETR_BENCH_ALT300=0 python3 bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['config']['streams'], d['config']['launch'], round(d['value'],1), 'images/s', round(d['ms_per_step'],3), 'ms')" || echo "streams $n failed"
done
