# On the GPU box: kernel-by-kernel timeline (start offset, duration, name) of ONE blocking 4 096-set batchVerify in latency mode (rocprofv3 --kernel-trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-ktrace_b4096}; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/tests/gpu_probe_aux.py ${WHAT:-b4096} > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last call: everything after the last k_blind
last = max(i for i, r in enumerate(rows) if "${ANCHOR:-k_blind}" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"]; n = n[n.find("k_"):] if "k_" in n else n
    print("%8.3f %8.3f  grid %6s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "?"), n[:60]))
PY
