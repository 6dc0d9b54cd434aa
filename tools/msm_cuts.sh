# On the GPU box: the 2^20-point G1 MSM under different window-group cuts (MI355_BLS_MSM_CUTS), two passes.
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
for cuts in ${CUTS:-8 10,4 11,5}; do
  echo -n "cuts=$cuts  "; MI355_BLS_MSM_CUTS=$cuts python3 tests/gpu_probe_aux.py msm 2>&1 | grep -o "'total': [0-9.]*\|msm two in flight: [0-9.]* ms" | tr '\n' ' '; echo
done; done
