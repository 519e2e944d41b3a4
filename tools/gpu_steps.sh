#!/bin/bash
# Run GPU steps one after another inside ONE gpurun call:  tools/gpu_steps.sh OUTDIR 'limit|name|command' ...
# An ordinary failure (exit 1: an assertion) lets the next step run; a step that is killed, times out or dies on a signal
# (exit >= 124) ends the call -- no further GPU step after a timeout or a fault.
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  limit=${spec%%|*}; rest=${spec#*|}; name=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${limit}s): $cmd" | tee -a "$out/steps.log"
  t0=$(date +%s)
  timeout -k 10 "$limit" bash -c "$cmd" > "$out/$name.log" 2> "$out/$name.err"
  rc=$?
  echo "=== $name exit $rc after $(( $(date +%s) - t0 )) s" | tee -a "$out/steps.log"
  tail -n 4 "$out/$name.log"
  if [ $rc -ge 124 ]; then echo "=== stopping: $name was killed / timed out / faulted" | tee -a "$out/steps.log"; exit $rc; fi
done
exit 0
