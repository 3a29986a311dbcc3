#!/bin/bash
# like run_exp.sh but C=1 at 1920x1280 (config-3-like single camera)
for n in "$@"; do
  if [ "$n" = base ]; then unset GSPLAT_MI355X_LIB; else export GSPLAT_MI355X_LIB=$PWD/scratch/exp/lib$n.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 20 --cams 1 --width 1920 --height 1280 --gaussians 3000000 > gpurun_out/exp1_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/exp1_$n.log; exit 1; }
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/exp1_{n}.log").read().strip().splitlines()[-1])
k=d["kernels_ms"]
print(f"{n:10s} C=1 3M 1920x1280: step {d['ms_per_step']:.3f} ms " + " ".join(f"{a.replace('gs_','')}={b:.3f}" for a,b in k.items()))
PY
done
