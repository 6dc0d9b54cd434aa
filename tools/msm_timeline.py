"""Timeline of the LAST G1 MSM call in a rocprofv3 --kernel-trace CSV: start / end of every kernel relative to the call's first kernel.
Usage: python3 tools/msm_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pip_convert" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
i0 = idx[which]
i1 = idx[which + 1] if which + 1 < len(idx) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    n = r["Kernel_Name"]
    n = n[n.find("k_"):][:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-30s %8.3f -> %8.3f  (%6.3f ms)  grid %s wg %s q %s" % (n, s, e, e - s, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r.get("Queue_Id", "?")))
