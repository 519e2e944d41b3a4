"""Launches per decoder layer, counted from a rocprofv3 kernel trace of an eager one-stream bench run:
the kernels between two consecutive rdetr::decoder_reference_kernel launches (one per layer) of the same step.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-extras --no-cpu-baseline   (RDETR_BENCH_STREAMS=1)
    python3 tools/count_decoder_launches.py OUT"""
import csv
import glob
import sys
from collections import Counter

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "decoder_reference_kernel" in n]
gaps = [b - a for a, b in zip(marks, marks[1:])]
print("launches between consecutive decoder_reference_kernel launches:", gaps)
layer = Counter(g for g in gaps if g < 60)
print("per decoder layer (layers 1-5: reference .. next reference):", dict(layer))
a, b = marks[1], marks[2]
for n in names[a:b]:
    print("   ", n[:110])
