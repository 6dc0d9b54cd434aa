# On the GPU box: SQ counters of the latency kernels (lane-team engine, row executor, tails), per launch; WHAT=b4096 (default) | fav | msm: the call of tests/gpu_probe_aux.py that is profiled
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-pmc_team}; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/a -o p -- python3 $R/tests/gpu_probe_aux.py ${WHAT:-b4096} > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/b -o p -- python3 $R/tests/gpu_probe_aux.py ${WHAT:-b4096} > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ("a", "b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not any(x in n for x in ("k_team", "k_hash_map", "k_tail", "k_hash_one", "k_pip_rowtail")): continue
            k = n[n.find("k_"):].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], d) not in seen and r["Counter_Name"] in ("SQ_WAVES", "GRBM_GUI_ACTIVE"):
                cnt[(k, d)] += 1; seen.add((r["Dispatch_Id"], d))
for k, c in sorted(acc.items()):
    la, lb = max(cnt[(k, "a")], 1), max(cnt[(k, "b")], 1)
    w = c["SQ_WAVES"] / la
    g = lambda name, l=la: c.get(name, 0.0) / l
    print("%-24s launches %d waves %6.0f  valu/wave %8.0f  lds/wave %7.0f salu/wave %7.0f vmem rd/wr per wave %5.0f/%5.0f  wavecyc/wave %9.0f  cyc/valu %.2f  wait_any %.3f wait_inst %.3f  lds: active %.3f wait %.3f bank_conflict/lds_active %.3f"
          % (k, la, w, g("SQ_INSTS_VALU") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_VMEM_RD", lb) / w, g("SQ_INSTS_VMEM_WR", lb) / w,
             4 * g("SQ_WAVE_CYCLES") / w, 4 * g("SQ_WAVE_CYCLES") / max(g("SQ_INSTS_VALU"), 1), g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
             g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_ACTIVE_INST_LDS", lb) / max(g("SQ_WAVE_CYCLES"), 1) , g("SQ_WAIT_INST_LDS", lb) / max(g("SQ_WAVE_CYCLES"), 1),
             g("SQ_LDS_BANK_CONFLICT", lb) / max(g("SQ_LDS_IDX_ACTIVE", lb), 1)))
PY
