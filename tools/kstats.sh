# per-kernel average durations of one-caller batches: rocprofv3 --kernel-trace --stats over bench.py --inflight 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-aux --inflight 1 > $O/log.txt 2>&1
rm -f $O/*kernel_trace.csv $O/*agent_info.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/s1_kernel_stats.csv")))[:14]:
    print(r["Name"][:48].ljust(48), r["Calls"], round(float(r["AverageNs"])/1e6,3))
PY
