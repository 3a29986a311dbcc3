#!/bin/bash
# usage: scratch/ab_lib3.sh name1 name2 ... -> bench each scratch/exp/libNAME.so ("base" = in-tree) twice, interleaved; rasterizer kernel times
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then unset GSPLAT_MI355X_LIB; else export GSPLAT_MI355X_LIB=$PWD/scratch/exp/lib$n.so; fi
  out=$(timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 30 $BENCH_ARGS 2>/dev/null) || { echo "$n FAILED"; exit 1; }
  python3 - "$n" <<PY
import json,sys
d=json.loads('''$out'''.strip().splitlines()[-1]); k=d["kernels_ms"]
print(f"{sys.argv[1]:14s} step {d['ms_per_step']:.4f} median {d['ms_per_step_median_hip_events']:.4f}  raster_fwd {k.get('gs_rasterize_fwd',0):.4f} raster_bwd {k.get('gs_rasterize_bwd',0):.4f}", flush=True)
PY
done; done
