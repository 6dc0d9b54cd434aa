# SQ / instruction-cache counters of prebuilt variants nim-blscurve_amd/variants/<name>.so, one caller (kernels alone), per kernel and launch.
# usage (GPU box): bash tools/pmc_ab.sh OUTDIR name1 name2 ...      -> OUTDIR/<name>.json + a table on stdout
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# a variant is selected with MI355_BLS_LIB (nim-blscurve_amd/__init__.py): the shipped library is never overwritten
P="$R/bench.py --no-cpu --no-aux --no-one-caller --steps 2 --warmup 1 --inflight 1 --ctx-mode throughput"
for v in "$@"; do
  export MI355_BLS_LIB=$R/nim-blscurve_amd/variants/$v.so
  rm -rf /tmp/pmc_$v; mkdir -p /tmp/pmc_$v
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH --output-format csv -d /tmp/pmc_$v/a -o p -- python3 $P > /tmp/pmc_$v/a.log 2>&1
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_$v/b -o p -- python3 $P > /tmp/pmc_$v/b.log 2>&1
  python3 $R/tools/summarize_pmc.py $O/$v.json /tmp/pmc_$v/a /tmp/pmc_$v/b > /dev/null 2>&1
done
unset MI355_BLS_LIB
python3 - "$O" "$@" <<'PY'
import json, sys
o, names = sys.argv[1], sys.argv[2:]
for k in ("k_hash_map", "k_hash_clear", "k_pkmul", "k_lines", "k_lineprod", "k_sig_bucket"):
    for n in names:
        try:
            e = json.load(open("%s/%s.json" % (o, n)))[k]
        except Exception as ex:
            print(k, n, "missing", ex); continue
        w = e.get("SQ_WAVES", 1)
        g = lambda c: e.get(c, 0.0)
        print("%-13s %-8s valu/wave %8.0f  wavecyc/wave %9.0f  kernel cyc %9.0f  cyc/valu %.2f  active_valu %.3f wait_any %.3f wait_inst %.3f  branches/wave %6.0f  icache req/wave %8.0f miss %.3f  ifetch_level/wavecyc %.3f" % (
            k, n, g("SQ_INSTS_VALU") / w, 4 * g("SQ_WAVE_CYCLES") / w, g("GRBM_GUI_ACTIVE") / 8, 4 * g("SQ_WAVE_CYCLES") / max(g("SQ_INSTS_VALU"), 1),
            g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
            g("SQ_INSTS_BRANCH") / w, g("SQC_ICACHE_REQ") / w, g("SQC_ICACHE_MISSES") / max(g("SQC_ICACHE_REQ"), 1), g("SQ_IFETCH_LEVEL") / max(g("SQ_WAVE_CYCLES"), 1)))
PY
