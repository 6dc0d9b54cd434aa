# timeline (start / end in ms, queue) of the kernels of the LAST 2^20-point MSM of tests/gpu_probe_aux.py msm (rocprofv3 --kernel-trace)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python3 $R/tests/gpu_probe_aux.py msm > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/kt/**/t_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last k_pip_convert marks the start of the last MSM
idx=[i for i,r in enumerate(rows) if "k_pip_convert" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0]
    print(n[:28].ljust(28), "q", r.get("Queue_Id","?").rjust(3), "%8.3f %8.3f" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6), "grid", r.get("Grid_Size_X", r.get("Grid_Size","?")))
PY
