R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in main exp; do
  rm -rf /tmp/kt_$v
  if [ $v = exp ]; then export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/exp.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o t -- python3 $R/tests/gpu_probe_aux.py fav > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
f=glob.glob("/tmp/kt_$v/**/t_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("k_hash_one","k_team_lines","k_tail","k_fold","k_lineprod","k_g1_sum","k_pip","k_key")): print(n[:50].ljust(50), r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
