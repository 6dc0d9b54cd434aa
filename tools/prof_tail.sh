# On the GPU box: k_tail / k_fold / k_hash_one durations of prebuilt variants (VARS="a b", nim-blscurve_amd/variants/<name>.so) under rocprofv3 --kernel-trace --stats,
# plus the launch-by-launch list of k_tail durations (tests/gpu_probe_aux.py fav b4096)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# a variant is selected with MI355_BLS_LIB (nim-blscurve_amd/__init__.py): the shipped library is never overwritten
for v in ${VARS:-g1dot rowe}; do
  export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/$v.so
  rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o t -- python3 $R/tests/gpu_probe_aux.py fav b4096 > /tmp/l.log 2>&1
  echo "== $v"; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pt/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ['k_tail','k_fold','k_hash_one','k_lines_coop','k_state_mul']):
        print(n[n.find('k_'):][:20], r['Calls'], round(float(r['AverageNs'])/1e6,3), round(int(r['MinNs'])/1e6,3), round(int(r['MaxNs'])/1e6,3))
f2=glob.glob('/tmp/pt/**/*kernel_trace.csv',recursive=True)[0]
print([round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3) for r in csv.DictReader(open(f2)) if 'k_tail' in r['Kernel_Name']])
PY
done
unset MI355_BLS_LIB
