#!/bin/bash
# Builds the CPU execution harness for the product's __host__ __device__ arithmetic (tests only).
set -e
cd "$(dirname "$0")"
mkdir -p _build
python3 ../../nim-blscurve_amd/tools/teamvm.py -o _build/teamvm_tables.inc.new      # the lane-team engine's programs, as the library embeds them
cmp -s _build/teamvm_tables.inc.new _build/teamvm_tables.inc || mv _build/teamvm_tables.inc.new _build/teamvm_tables.inc
if [ ! -f _build/libemu.so ] || [ _build/teamvm_tables.inc -nt _build/libemu.so ] || [ emu.hip -nt _build/libemu.so ] || [ -n "$(find ../../nim-blscurve_amd/csrc -name '*.hpp' -newer _build/libemu.so)" ]; then
  hipcc -O2 -std=c++17 --offload-host-only -DBLS_TRACK_BOUNDS -g -rdynamic -fPIC -shared -I ../../nim-blscurve_amd/csrc emu.hip -o _build/libemu.so
fi
