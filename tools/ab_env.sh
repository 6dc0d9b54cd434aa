# On the GPU box: same-library A/B of an experiment knob on the latency rows of bench.py: VARIANT's library with and without `KNOB=1`, alternating.
#   gpurun -- 'VARIANT=exp KNOB=MI355_BLS_HASH_MAP_NO_ROWS bash tools/ab_env.sh'
R=$GRAFT_REPO_ROOT; export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/${VARIANT:?}.so
one() { python3 $R/bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | head -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); a=d['aux']
print(round(d['ms_per_step'],3), 'fav', round(a['fastAggregateVerify_32768']['ms_per_call'],3), 'one', round(a['verify_one_signature']['ms_per_call'],3), 'b64', round(a['batchVerify_64']['ms_per_blocking_call'],3), 'b4096', round(a['batchVerify_4096']['ms_per_blocking_call'],3), 'curve', [round(r['ms_per_blocking_call'],3) for r in a['latency_curve'][:7]])"; }
for r in 1 2 3; do
  echo -n "default      "; one
  echo -n "${KNOB}=1  "; env ${KNOB}=1 python3 -c "pass"; eval "${KNOB}=1 one"
done
python3 -m pytest $R/tests/test_gpu_clear_chain.py $R/tests/test_gpu_fav.py $R/tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3
