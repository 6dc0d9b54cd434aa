R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats_msm; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o m -- python3 $R/tests/gpu_probe_msm.py > $O/log.txt 2>&1
rm -f $O/*kernel_trace.csv $O/*agent_info.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/m_kernel_stats.csv")))[:16]:
    print(r["Name"].split("::")[-1].split("(")[0][:40].ljust(40), r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
