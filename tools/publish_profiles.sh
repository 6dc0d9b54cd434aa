# In the build container, after `TAG=rNN bash tools/collect_profiles.sh` ran on the GPU box and gpurun merged gpurun_out/rNN back: copy the evidence the
# judge reads into profiles/ under the round's names.   bash tools/publish_profiles.sh r06 [dir with gpu_tests.txt / soak.txt]
T=${1:?round tag}; S=gpurun_out/$T; X=${2:-}
cp $S/bench.json profiles/${T}_bench.json
cp $S/stats3/s3_kernel_stats.csv profiles/${T}_kernel_stats_default_inflight.csv
cp $S/stats1/s1_kernel_stats.csv profiles/${T}_kernel_stats_1inflight.csv
cp $S/statsL/sL_kernel_stats.csv profiles/${T}_kernel_stats_latency_mode.csv
cp $S/aux/aux_kernel_stats.csv profiles/${T}_kernel_stats_aux.csv
cp $S/pmc_summary.json profiles/${T}_pmc_summary.json
cp $S/pmc_summary_msm.json profiles/${T}_pmc_summary_msm.json
[ -f $S/kernel_metadata.txt ] && cp $S/kernel_metadata.txt profiles/${T}_kernel_metadata.txt
if [ -n "$X" ]; then cp $X/gpu_tests.txt profiles/${T}_gpu_tests.txt; cp $X/soak.txt profiles/${T}_soak.txt; fi
ls -la profiles/${T}_*
