# On the GPU box: the 2^20-point G1 MSM with the segment reduction on lane teams of MI355_BLS_MSM_TEAM = 1 (none), 2, 4 lanes; two passes.
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
for t in ${TEAMS:-1 2 4}; do
  echo -n "team=$t  "; MI355_BLS_MSM_TEAM=$t python3 tests/gpu_probe_aux.py msm 2>&1 | grep -o "'total': [0-9.]*\|msm two in flight: [0-9.]* ms" | tr '\n' ' '; echo
done; done
