#!/bin/bash
# usage: scratch/build_exp.sh NAME "-DGS_EXP=1 ..."   -> scratch/exp/libNAME.so (rasterize.hip variant, other objects from build/)
set -e
cd /root/repo/3d-reconstruction_amd/csrc
mkdir -p ../../scratch/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $2 -c rasterize.hip -o ../../scratch/exp/rasterize_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../build/projection.o ../build/sh.o ../build/isect.o ../build/isect_bin.o ../build/radix_sort.o ../build/api.o ../../scratch/exp/rasterize_$1.o -o ../../scratch/exp/lib$1.so
