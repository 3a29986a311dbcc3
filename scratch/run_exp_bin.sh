#!/bin/bash
# usage: scratch/run_exp_bin.sh "name slices" ...   (slices = total workgroups over all cameras; 0 = default)
for spec in "$@"; do
  set -- $spec; n=$1; sl=$2
  if [ "$n" = base ]; then unset GSPLAT_MI355X_LIB; else export GSPLAT_MI355X_LIB=$PWD/scratch/exp/lib$n.so; fi
  if [ "$sl" = 0 ]; then unset GS_BIN_SLICES; else export GS_BIN_SLICES=$sl; fi
  for shape in "" "--cams 1 --width 1920 --height 1280 --gaussians 3000000"; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 $shape > gpurun_out/expb.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/expb.log; exit 1; }
    python - "$n/$sl" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/expb.log").read().strip().splitlines()[-1])
k=d["kernels_ms"]
print(f"{sys.argv[1]:14s} C={d['config']['cams_per_gpu']} step {d['ms_per_step']:.3f} ms  count {k['gs_isect_bin_count']:.4f} offs {k['gs_isect_bin_offsets']:.4f} scat {k['gs_isect_bin_scatter']:.4f} sort {k['gs_isect_tile_sort']:.4f}")
PY
  done
done
