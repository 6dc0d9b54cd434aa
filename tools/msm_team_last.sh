# On the GPU box (experiment build: BLS_EXTRA_FLAGS=-DBLS_EXPERIMENTS BLS_OUT=variants/exp.so nim-blscurve_amd/build.sh): lane teams for the segment sums of the
# LAST window group only (nothing else runs beside it), by segment length.  Three runs each; "msm {...}" = stage times of one call, then two in flight.
R=$GRAFT_REPO_ROOT; export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/exp.so
for rep in 1 2 3; do
for cfg in "0 0" "2 0" "4 0" "2 8" "4 8" "0 8" "4 32"; do
  set -- $cfg
  echo -n "team_last=$1 seg=$2: "
  MI355_BLS_MSM_TEAM_LAST=$1 MI355_BLS_MSM_SEG=$2 python3 $R/tests/gpu_probe_aux.py msm 2>/dev/null | python3 -c "
import sys,ast
l=sys.stdin.read().splitlines()
d=ast.literal_eval(l[0][4:]); print(round(d['total'],3), 'sig_mul_sum(last reduction)', round(d['sig_mul_sum'],3), '|', l[1])"
done; done
