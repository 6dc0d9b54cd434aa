# k_tail / k_fold average durations (rocprofv3 --kernel-trace --stats of tests/gpu_probe_lat.py) for prebuilt variants/<name>.so
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
# a variant is selected with MI355_BLS_LIB (nim-blscurve_amd/__init__.py): the shipped library is never overwritten
for v in "$@"; do
  export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/$v.so
  rm -rf /tmp/kt_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o t -- python3 $R/tests/gpu_probe_lat.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/kt_$v/**/t_kernel_stats.csv",recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for k in ("k_tail","k_fold","k_lineprod2"):
        if k+"(" in n: out.append("%s calls %s avg %.3f min %.3f" % (k, r["Calls"], float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6))
print("$v:", "; ".join(out))
PY
done
unset MI355_BLS_LIB
