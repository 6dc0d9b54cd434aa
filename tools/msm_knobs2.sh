# On the GPU box (experiment build in variants/exp.so): round-6 MSM knobs, alternating: default (row tail, conversion beside the sort, two window groups),
# conversion in line, one window group.
R=$GRAFT_REPO_ROOT; export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/exp.so
one() { python3 $R/tests/gpu_probe_aux.py msm 2>/dev/null | python3 -c "
import sys,ast
l=sys.stdin.read().splitlines()
d=ast.literal_eval(l[0][4:]); print(round(d['total'],3), {k:round(v,3) for k,v in d.items() if v and k!='total'}, '|', l[1])"; }
for rep in 1 2 3 4; do
  echo -n "default      "; one
  echo -n "conv inline  "; MI355_BLS_MSM_CONV_INLINE=1 one
  echo -n "one group    "; MI355_BLS_MSM_NOSPLIT=1 one
  echo -n "one group s8 "; MI355_BLS_MSM_NOSPLIT=1 MI355_BLS_MSM_SEG=8 one
done
python3 -m pytest $R/tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -2
