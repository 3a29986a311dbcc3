#!/bin/bash
# usage: scratch/run_exp.sh name1 name2 ...  -> bench each scratch/exp/libNAME.so ("base" = in-tree lib), print raster kernel ms
for n in "$@"; do
  if [ "$n" = base ]; then unset GSPLAT_MI355X_LIB; else export GSPLAT_MI355X_LIB=$PWD/scratch/exp/lib$n.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/exp_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/exp_$n.log; exit 1; }
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/exp_{n}.log").read().strip().splitlines()[-1])
k=d["kernels_ms"]
print(f"{n:10s} step {d['ms_per_step']:.3f} ms  fwd {k['gs_rasterize_fwd']:.4f}  bwd {k['gs_rasterize_bwd']:.4f}  projf {k['gs_projection_fwd']:.4f} projb {k['gs_projection_bwd']:.4f} sort {k['gs_isect_tile_sort']:.4f} scat {k['gs_isect_bin_scatter']:.4f}")
PY
done
