# The evidence under profiles/ (round tag $TAG, default r04).  On the GPU box: bash tools/collect_profiles.sh
R=$GRAFT_REPO_ROOT; TAG=${TAG:-r05}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu --no-aux --no-one-caller"
# kernel-trace statistics: the default bench command (3 chained batches in flight), the same kernels alone (one throughput-mode
# caller), one latency-mode caller, and the aux configs (MSM 2^20, fastAggregateVerify 32768, 4096-tuple batch)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o s3 -- $B --steps 5 --warmup 1 > $O/stats3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o s1 -- $B --steps 5 --warmup 1 --inflight 1 --ctx-mode throughput > $O/stats1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsL -o sL -- $B --steps 5 --warmup 1 --inflight 1 --ctx-mode latency > $O/statsL.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/aux -o aux -- python3 $R/tests/gpu_probe_aux.py > $O/aux.log 2>&1
# counters: separate passes, kernel-trace/stats only (never with sys/hip/hsa tracing)
P="$B --steps 2 --warmup 1 --inflight 1 --ctx-mode throughput"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $P > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $P > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- $P > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT --output-format csv -d $O/pmc_sq2 -o p -- $P > $O/pmc_sq2.log 2>&1
M="python3 $R/tests/gpu_probe_aux.py msm"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_aux -o p -- $M > $O/pmc_fetch_aux.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_aux -o p -- $M > $O/pmc_write_aux.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq_aux -o p -- $M > $O/pmc_sq_aux.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT --output-format csv -d $O/pmc_sq2_aux -o p -- $M > $O/pmc_sq2_aux.log 2>&1
python3 $R/tools/summarize_pmc.py $O/pmc_summary.json $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.log 2>&1
python3 $R/tools/summarize_pmc.py $O/pmc_summary_msm.json $O/pmc_fetch_aux $O/pmc_write_aux $O/pmc_sq_aux $O/pmc_sq2_aux >> $O/pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
# the plain bench line of the same build
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
du -sh $O; find $O -type f | head -40; tail -c 300 $O/stats3.log
