# MSM 2^20 on the GPU box: parity tests, the stage timers of six runs, and the per-kernel averages (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
python3 -m pytest $R/tests/test_gpu_msm.py $R/tests/test_gpu_combine.py -m gpu -x -q 2>&1 | tail -3
python3 $R/tests/gpu_probe_aux.py msm 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o aux -- python3 $R/tests/gpu_probe_aux.py msm > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/kp/**/aux_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    n=r["Name"].replace("(anonymous namespace)::","")
    print(n[:44].ljust(44), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3))
PY
