# CPU only: the product's __host__ __device__ arithmetic (fields, curves, hash-to-G2, pairing, the Fp12 engine's emulation, the bounds tracker)
# built with UndefinedBehaviorSanitizer (signed overflow and shifts included, no recovery) and run through tests/test_host_emu.py.
# GPU sanitizers are not available on this pool; this is the sanitizer run of the same source on the host.   bash tools/ubsan_host_emu.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); B=$R/tests/host_emu/_build; RT=$(dirname $(find /opt/rocm/lib/llvm -name "libclang_rt.ubsan_standalone-x86_64.so" | head -1))
mkdir -p $B
hipcc -O1 -std=c++17 --offload-host-only -DBLS_TRACK_BOUNDS -g -fPIC -shared -fsanitize=undefined,signed-integer-overflow,shift -fno-sanitize-recover=undefined \
  -shared-libsan -Wno-option-ignored -I $R/nim-blscurve_amd/csrc $R/tests/host_emu/emu.hip -o $B/libemu_ubsan.so
[ -f $B/libemu.so ] && cp $B/libemu.so $B/libemu_plain.so
trap '[ -f $B/libemu_plain.so ] && mv $B/libemu_plain.so $B/libemu.so; touch $B/libemu.so' EXIT
cp $B/libemu_ubsan.so $B/libemu.so; touch $B/libemu.so
cd $R && LD_LIBRARY_PATH=$RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_host_emu.py -x -q 2>&1 | tail -3
