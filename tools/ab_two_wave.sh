# On the GPU box: same-library A/B of k_hash_clear at two waves per SIMD (variants/two.so, built with -DBLS_CLEAR_TWO_WAVE -DBLS_EXPERIMENTS) against
# the one-wave form (MI355_BLS_CLEAR_ONE_WAVE=1 makes the same library launch it): pipelined step, one caller, kernels alone; then the parity tests
# at the headline sizes on the two-wave kernel.
R=$GRAFT_REPO_ROOT; export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/two.so
one() { python3 $R/bench.py --steps 30 --warmup 4 --no-cpu --no-aux ${EXTRA} 2>/dev/null | head -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['ms_one_caller'],2), {k:round(v,3) for k,v in d['kernel_ms_alone'].items()}, 'ceiling GHz', d['roofline']['ceiling_clock_ghz'])"; }
for r in 1 2 3; do
  echo -n "two-wave  "; one
  echo -n "one-wave  "; MI355_BLS_CLEAR_ONE_WAVE=1 one
done
echo "131072 tuples per batch:"
for r in 1 2; do
  echo -n "two-wave  "; EXTRA="--batch 131072 --no-one-caller" one 2>/dev/null || true
  echo -n "one-wave  "; EXTRA="--batch 131072 --no-one-caller" MI355_BLS_CLEAR_ONE_WAVE=1 one 2>/dev/null || true
done
python3 -m pytest $R/tests/test_gpu_headline.py -x -q -m gpu -k "not 2pow20" 2>&1 | tail -3
