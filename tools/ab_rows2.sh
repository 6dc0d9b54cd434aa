# On the GPU box (experiment build in variants/exp.so): the row executor at two workgroups per CU for 225 .. 448 items (MI355_BLS_ROWVM2=1) against the
# lane-team engine, sizes between the points of bench.py's latency curve; then the new hand-over tests on the variant with the knob on.
R=$GRAFT_REPO_ROOT; export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/exp.so
for r in 1 2; do
  echo "== lane-team engine above 224"; LAT_SIZES=200,240,300,400,448,512 python3 $R/tests/gpu_probe_lat.py 2>/dev/null | grep "^batch" | cut -c1-150
  echo "== rows, two workgroups per CU"; MI355_BLS_ROWVM2=1 LAT_SIZES=200,240,300,400,448,512 python3 $R/tests/gpu_probe_lat.py 2>/dev/null | grep "^batch" | cut -c1-150
done
MI355_BLS_ROWVM2=1 python3 -m pytest $R/tests/test_gpu_batch.py -x -q -m gpu -k "hand_over or fold" 2>&1 | tail -3
