# SQ / HBM counters of the Pippenger kernels at 2^20 points (tests/gpu_probe_aux.py msm), separate --pmc passes -> $O/pmc_summary_msm.json
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r03}; O=$R/gpurun_out/${TAG}_msm; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P="python3 $R/tests/gpu_probe_aux.py msm"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o aux -- $P > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $P > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $P > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $P > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT --output-format csv -d $O/pmc_sq2 -o p -- $P > $O/pmc_sq2.log 2>&1
python3 $R/tools/summarize_pmc.py $O/pmc_summary_msm.json $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
python3 - <<PY
import json
d=json.load(open("$O/pmc_summary_msm.json"))
for k,v in sorted(d.items()):
    if k.startswith("k_pip") or k.startswith("k_msm"):
        print(k.ljust(22), {a: round(b,3) for a,b in v.items() if a in ("cycles_per_valu","wait_any_share","mad_share_of_valu","SQ_WAVES","SQ_INSTS_VALU","hbm_bytes_raw")})
PY
