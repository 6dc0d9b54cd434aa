R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for k in 2 3; do
rocprofv3 --kernel-trace --output-format csv -d $O/t$k -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-aux --inflight $k > $O/log$k.txt 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/t$k/t_kernel_trace.csv")))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("::")[-1].split("(")[0],r["Queue_Id"],r.get("Stream_Id","")) for r in rows if "::k_" in r["Kernel_Name"] and "k_sign" not in r["Kernel_Name"]]
ev.sort()
t0=ev[0][0]
# last 40% of the run
cut=ev[int(len(ev)*0.5)][0]
sel=[e for e in ev if e[0]>=cut]
for e in sel[:60]:
    print(round((e[0]-cut)/1e6,2), round((e[1]-cut)/1e6,2), e[2], "q"+e[3], e[4])
PY
done
rm -rf $O/t2 $O/t3
