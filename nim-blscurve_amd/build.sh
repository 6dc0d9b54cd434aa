#!/bin/bash
# Builds the C-ABI shared library (HIP kernels for gfx950) in-tree.
set -e
cd "$(dirname "$0")"
OUT=libblscurve_mi355x.so
if [ "$1" != "-f" ] && [ -f $OUT ] && [ -z "$(find csrc ../include -newer $OUT -type f)" ]; then
  exit 0
fi
# --gpu-max-threads-per-block=64: every kernel is one wave per workgroup; this also gives the out-of-line device
# functions the full 512-register (VGPR+AGPR) budget instead of the 128-VGPR default, so they stop spilling to scratch
hipcc -O3 -std=c++17 --offload-arch=gfx950 --gpu-max-threads-per-block=64 -fPIC -shared csrc/kernels.hip -o $OUT.tmp
mv $OUT.tmp $OUT
