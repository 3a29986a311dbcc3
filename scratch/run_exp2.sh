#!/bin/bash
# usage: BENCH_ARGS="--sh-backward split" scratch/run_exp2.sh name1 name2 ...  -> bench each scratch/exp/libNAME.so ("base" = in-tree)
for n in "$@"; do
  if [ "$n" = base ]; then unset GSPLAT_MI355X_LIB; else export GSPLAT_MI355X_LIB=$PWD/scratch/exp/lib$n.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 $BENCH_ARGS > gpurun_out/exp_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/exp_$n.log; exit 1; }
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/exp_{n}.log").read().strip().splitlines()[-1])
k=d["kernels_ms"]
print(f"{n:10s} step {d['ms_per_step']:.3f} ms (median {d['ms_per_step_median_hip_events']:.3f})  " + "  ".join(f"{a.replace('gs_','')} {b:.4f}" for a,b in k.items() if any(s in a for s in ('raster',))), flush=True)
PY
done
