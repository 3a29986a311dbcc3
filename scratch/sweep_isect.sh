#!/bin/bash
# launch-shape sweep of the count / scatter passes (env overrides read by bench.py)
run() {
  out=$(env "$@" timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 20 $EXTRA 2>/dev/null) || { echo "$* FAILED"; exit 1; }
  python3 - "$*" <<PY
import json,sys
d=json.loads('''$out'''.strip().splitlines()[-1]); k=d["kernels_ms"]
print(f"{sys.argv[1]:60s} step {d['ms_per_step']:.3f} count {k['gs_isect_bin_count']:.4f} offsets {k['gs_isect_bin_offsets']:.4f} scatter {k['gs_isect_bin_scatter']:.4f} sort {k['gs_isect_tile_sort']:.4f}", flush=True)
PY
}
run GS_SLICES_TOTAL=512
for t in 256 384 768 1024; do run GS_SLICES_TOTAL=$t; done
for r in 1 3 4; do run GS_SCATTER_REGIONS=$r; done
for w in 128 512 1024; do run GS_SCATTER_WORKGROUPS=$w; done
run GS_SLICES_TOTAL=512
