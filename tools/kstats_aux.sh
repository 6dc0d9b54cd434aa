# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the aux configs: MSM 2^20, fastAggregateVerify 32768, 4096-tuple batches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats_aux; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o aux -- python3 $R/tests/gpu_probe_aux.py > $O/log.txt 2>&1
rm -f $O/*kernel_trace.csv $O/*agent_info.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/aux_kernel_stats.csv")))[:40]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e6,3), round(float(r["TotalDurationNs"])/1e6,2))
PY
