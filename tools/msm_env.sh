# On the GPU box: the 2^20-point G1 MSM under environment settings given as arguments ("A=1 B=2" each), two passes.
cd $GRAFT_REPO_ROOT
for pass in 1 2 3; do
for e in "$@"; do
  echo -n "$e  "; env $e python3 tests/gpu_probe_aux.py msm 2>&1 | grep -o "'total': [0-9.]*\|msm two in flight: [0-9.]* ms" | tr '\n' ' '; echo
done; done
