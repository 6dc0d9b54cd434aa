# On the GPU box: same-box A/B of a variant library against the main one on the latency rows of bench.py (aux configs), alternating, three times.
#   gpurun -- 'VARIANT=pin bash tools/ab_lib.sh > gpurun_out/ab.txt'
R=$GRAFT_REPO_ROOT; V=$R/nim-blscurve_amd/variants/${VARIANT:?}.so
one() { python3 $R/bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | head -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); a=d['aux']
print(round(d['ms_per_step'],3), 'fav', a['fastAggregateVerify_32768']['ms_per_call'], 'one', a['verify_one_signature']['ms_per_call'], 'b64', a['batchVerify_64']['ms_per_blocking_call'], 'b4096', a['batchVerify_4096']['ms_per_blocking_call'], 'curve', [round(r['ms_per_blocking_call'],3) for r in a['latency_curve'][:7]], 'fav stages', a['fastAggregateVerify_32768'].get('stage_ms'))"; }
for r in 1 2 3; do
  echo -n "main     "; one
  echo -n "$VARIANT  "; MI355_BLS_LIB=$V one
done
MI355_BLS_LIB=$V python3 -m pytest $R/tests/test_gpu_clear_chain.py $R/tests/test_gpu_fav.py $R/tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3
