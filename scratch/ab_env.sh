#!/bin/bash
# usage: BENCH_ARGS="..." scratch/ab_env.sh VAR val1 val2 ...  -> bench with GS_<VAR>=val, interleaved, three rounds
var=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
  out=$(env GS_$var=$v timeout -k 10 150 python3 bench.py --no-cpu-baseline --steps 50 $BENCH_ARGS 2>/dev/null) || { echo "$v FAILED"; exit 1; }
  python3 - "$var=$v" <<PY
import json,sys
d=json.loads('''$out'''.strip().splitlines()[-1])
print(f"{sys.argv[1]:22s} step {d['ms_per_step']:.4f} median {d['ms_per_step_median_hip_events']:.4f}", flush=True)
PY
done; done
