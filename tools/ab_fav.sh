# On the GPU box: stage times of fastAggregateVerify(32 768) (tests/gpu_probe_aux.py fav, sixth call) for the main library and a variant, alternating.
R=$GRAFT_REPO_ROOT; V=$R/nim-blscurve_amd/variants/${VARIANT:?}.so
for r in 1 2 3; do
  echo -n "main     "; python3 $R/tests/gpu_probe_aux.py fav 2>/dev/null | tail -1
  echo -n "$VARIANT  "; MI355_BLS_LIB=$V python3 $R/tests/gpu_probe_aux.py fav 2>/dev/null | tail -1
done
