# On the GPU box (experiment build in variants/exp.so): the G1 MSM with its tail on the row arithmetic (k_pip_rowtail) against the one-lane window sums
# (MI355_BLS_MSM_NO_ROWTAIL=1), alternating; then the MSM / combine parity tests on the new tail.
R=$GRAFT_REPO_ROOT; export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/exp.so
one() { python3 $R/tests/gpu_probe_aux.py msm 2>/dev/null | python3 -c "
import sys,ast
l=sys.stdin.read().splitlines()
d=ast.literal_eval(l[0][4:]); print(round(d['total'],3), {k:round(v,3) for k,v in d.items() if v and k!='total'}, '|', l[1])"; }
for rep in 1 2 3; do
  echo -n "rowtail   "; one
  echo -n "one-lane  "; MI355_BLS_MSM_NO_ROWTAIL=1 one
done
for lg in 14 16 18; do
  echo -n "2^$lg rowtail   "; MSM_LOG2=$lg one
  echo -n "2^$lg one-lane  "; MSM_LOG2=$lg MI355_BLS_MSM_NO_ROWTAIL=1 one
done
python3 -m pytest $R/tests -x -q -m gpu -k "msm or combine or pippenger" 2>&1 | tail -4
