#!/bin/bash
# usage: EXTRA="--mode composite" scratch/ab_isect.sh "ENV=VAL ..." ...  -> bench each setting twice, interleaved, isect kernel times
for rep in 1 2; do
for e in "$@"; do
  out=$(env $e timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 20 $EXTRA 2>/dev/null) || { echo "$e FAILED"; exit 1; }
  python3 - "$e" <<PY
import json,sys
d=json.loads('''$out'''.strip().splitlines()[-1])
k=d["kernels_ms"]
print(f"{sys.argv[1]:50s} step {d['ms_per_step']:.4f} count {k.get('gs_isect_bin_count',0):.4f} scatter {k.get('gs_isect_bin_scatter',0):.4f} sort {k.get('gs_isect_tile_sort',0):.4f}", flush=True)
PY
done; done
