#!/bin/bash
# Builds the C-ABI shared library (HIP kernels for gfx950) in-tree.
#
#   device side:  hipcc -S  ->  tools/align_isa.py (every 8-byte instruction 8-byte aligned, see its header)
#                 ->  assemble  ->  link the code object  ->  bundle
#   host side:    hipcc --cuda-host-only with the bundle embedded (-fcuda-include-gpubinary), linked to libamdhip64
# This is what `hipcc -shared` does in one go (see `hipcc -###`), with the post-pass spliced in between.  If any
# step of the spliced pipeline fails the BUILD FAILS: the plain one-step hipcc build (same code, unaligned, ~23 % slower issue of
# 8-byte VALU streams) is made only when BLS_NO_ALIGN=1 asks for it, and the library says which one it is (mi355_bls_build_info).
set -e
cd "$(dirname "$0")"
OUT=${BLS_OUT:-libblscurve_mi355x.so}
# up to date = the library was built from exactly these sources (content hash, not time stamps: a copy of the tree need not keep them)
STAMP=$(cat csrc/* ../include/*.h tools/align_isa.py tools/gen_lineprod_asm.py tools/asmlib.py tools/gen_lines_asm.py tools/gen_clear_asm.py tools/gen_msm_asm.py tools/gen_pkmul_asm.py tools/gen_pow_asm.py tools/teamvm.py build.sh | sha256sum | cut -d" " -f1)-$BLS_EXTRA_FLAGS-$BLS_NO_ALIGN
if [ "$1" != "-f" ] && [ -f $OUT ] && [ "$(cat $OUT.stamp 2>/dev/null)" = "$STAMP" ]; then
  exit 0
fi
LLVM=/opt/rocm/lib/llvm/bin
# --gpu-max-threads-per-block=64: the default launch bound (most kernels are one wave per workgroup; the Fp12 engine kernels and the LDS sort declare their own); this also gives the out-of-line device
# functions the full 512-register (VGPR+AGPR) budget instead of the 128-VGPR default, so they stop spilling to scratch
# -amdgpu-dpp-combine=false: on gfx950 the "rev" VOP2 opcodes (v_subrev_u32, v_lshlrev_b32 ...) apply a DPP lane permutation to src1, not to src0 as the
# ISA documents and as LLVM's DPP combine assumes when it folds a v_mov_b32_dpp into a non-commutative consumer (measured: tools/test_dpp.hip;
# found through a wrong a - b behind a quad_perm exchange in the G1 lane teams).  The lane exchanges stay plain v_mov_b32_dpp.
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 --gpu-max-threads-per-block=64 -mllvm -amdgpu-dpp-combine=false $BLS_EXTRA_FLAGS"      # BLS_EXTRA_FLAGS: -D switches of A/B experiments (tools/abn.sh)
B=build
mkdir -p $B
python3 tools/gen_lineprod_asm.py -o $B/lineprod_asm.inc         # the hand-allocated inner loop of k_lineprod (included by csrc/kernels.hip)
python3 tools/gen_lines_asm.py -o $B/lines_asm.inc               # the 68-step Miller walk of k_lines (tools/asmlib.py)
python3 tools/gen_clear_asm.py -o $B/clear_asm.inc               # k_hash_clear's body (cofactor clearing of hash-to-G2)
python3 tools/gen_clear_asm.py --two-wave -o $B/clear2_asm.inc   # the same body for 256 registers (experiment builds with -DBLS_CLEAR_TWO_WAVE only)
python3 tools/gen_msm_asm.py -o $B/msm_asm.inc                   # the bucket accumulation of the G1 Pippenger MSM
python3 tools/gen_pkmul_asm.py -o $B/pkmul_asm.inc               # [r]PK of the batch path
python3 tools/gen_pow_asm.py -o $B/pow_asm.inc                   # a^((p-3)/4): the exponentiation behind every square root (hash-to-G2's SSWU maps, decompression)
python3 tools/teamvm.py -o $B/teamvm_tables.inc                  # programs of the lane-team engine (csrc/teamvm.hpp): cofactor clearing, Miller walk
aligned_build() {
  hipcc $FLAGS --cuda-device-only -S -o $B/dev.s csrc/kernels.hip || return 1
  python3 tools/align_isa.py $B/dev.s $B/dev_aligned.s --clang $LLVM/clang --objdump $LLVM/llvm-objdump || return 1
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $B/dev_aligned.s -o $B/dev.o || return 1
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $B/dev.co $B/dev.o || return 1
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input=$B/dev.co -output=$B/dev.hipfb || return 1
  hipcc $FLAGS -DBLS_BUILD_ALIGNED=1 "-DBLS_BUILD_STAMP=\"${STAMP%%-*}\"" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $B/dev.hipfb -fPIC -shared csrc/kernels.hip -o $OUT.tmp || return 1
}
if [ "$BLS_NO_ALIGN" = "1" ]; then
  echo "build.sh: BLS_NO_ALIGN=1: plain hipcc build (no alignment post-pass)" >&2
  hipcc $FLAGS -DBLS_BUILD_ALIGNED=0 "-DBLS_BUILD_STAMP=\"${STAMP%%-*}\"" -fPIC -shared csrc/kernels.hip -o $OUT.tmp
elif ! aligned_build; then
  echo "build.sh: the aligned build failed (see above); set BLS_NO_ALIGN=1 to build without the alignment post-pass" >&2
  exit 1
fi
mv $OUT.tmp $OUT
# register / spill / scratch / LDS table of every kernel, beside the library (it travels with it; bench.py reports the spill counts of the kernels it ran)
if [ -f $B/dev_aligned.s ] && [ "$BLS_NO_ALIGN" != "1" ]; then
  python3 ../tools/kernel_metadata.py --json $B/dev_aligned.s > $OUT.kmeta.json || rm -f $OUT.kmeta.json
fi
echo "$STAMP" > $OUT.stamp
