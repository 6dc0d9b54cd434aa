R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o s3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-aux > $O/stats3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o s1 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-aux --inflight 1 > $O/stats1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-aux --inflight 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-aux --inflight 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-aux --inflight 1 > $O/pmc_sq.log 2>&1
rm -f $O/*/*kernel_trace.csv $O/*/*agent_info.csv
du -sh $O; ls $O/*; tail -1 $O/stats3.log | cut -c1-200
