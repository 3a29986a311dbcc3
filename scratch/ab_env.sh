#!/bin/bash
# usage: scratch/ab_env.sh "ENV=VAL ..." "ENV=VAL ..." ...   -> bench each setting twice, interleaved
for rep in 1 2; do
for e in "$@"; do
  out=$(env $e timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 30 $BENCH_ARGS 2>/dev/null) || { echo "$e FAILED"; exit 1; }
  python3 - "$e" <<PY
import json,sys
d=json.loads('''$out'''.strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} step {d['ms_per_step']:.4f} median {d['ms_per_step_median_hip_events']:.4f}", flush=True)
PY
done; done
