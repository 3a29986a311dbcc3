#!/bin/bash
# usage: scratch/build_exp.sh NAME FILE "-DGS_EXP=1 ..."  -> scratch/exp/libNAME.so (FILE.hip variant, other objects from build/)
set -e
cd /root/repo/3d-reconstruction_amd/csrc
mkdir -p ../../scratch/exp
EXTRA=""; [ "$2" = rasterize ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA $3 -c $2.hip -o ../../scratch/exp/$2_$1.o
OBJS=""
for f in projection sh isect isect_bin radix_sort rasterize densify activate rigid refine loss cubemap api; do
  if [ $f = $2 ]; then OBJS="$OBJS ../../scratch/exp/$2_$1.o"; else OBJS="$OBJS ../build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../../scratch/exp/lib$1.so
