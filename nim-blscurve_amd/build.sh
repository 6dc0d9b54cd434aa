#!/bin/bash
# Builds the C-ABI shared library (HIP kernels for gfx950) in-tree.
set -e
cd "$(dirname "$0")"
OUT=libblscurve_mi355x.so
if [ "$1" != "-f" ] && [ -f $OUT ] && [ -z "$(find csrc ../include -newer $OUT -type f)" ]; then
  exit 0
fi
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared csrc/kernels.hip -o $OUT.tmp
mv $OUT.tmp $OUT
