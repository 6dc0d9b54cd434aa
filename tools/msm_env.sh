# On the GPU box: the 2^20-point G1 MSM under environment settings given as arguments ("A=1 B=2" each; "-" = none), three passes.  The experiment
# switches (MI355_BLS_MSM_*) exist only in a build made with BLS_EXTRA_FLAGS=-DBLS_EXPERIMENTS: LIB=<name> selects nim-blscurve_amd/variants/<name>.so
cd $GRAFT_REPO_ROOT
[ -n "$LIB" ] && export MI355_BLS_LIB=$GRAFT_REPO_ROOT/nim-blscurve_amd/variants/$LIB.so
for pass in 1 2 3; do
for e in "$@"; do
  [ "$e" = "-" ] && e="X_NONE=1"
  echo -n "$e  "; env $e python3 tests/gpu_probe_aux.py msm 2>&1 | grep -o "'total': [0-9.]*\|msm two in flight: [0-9.]* ms" | tr '\n' ' '; echo
done; done
