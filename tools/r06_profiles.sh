# round-6 evidence pass (GPU box): kernel stats, in-graph timelines and PMC summaries of the shipped library -> gpurun_out/r06_*
R=$GRAFT_REPO_ROOT
cd $R
for CFG in cfg2 cfg4 cfg5; do
  BENCH_ARGS="--config $CFG" bash profiles/kt_pass.sh r06_$CFG; mv gpurun_out/kt_r06_$CFG.txt gpurun_out/r06_${CFG}_kernel_stats.txt
  BENCH_ARGS="--config $CFG" bash tools/run_timeline2.sh r06_$CFG --config $CFG; mv gpurun_out/tl_r06_$CFG.txt gpurun_out/r06_${CFG}_graph_timeline.txt
done
bash tools/run_timeline2.sh r06_cfg2_dense --input dense --block1 generic; mv gpurun_out/tl_r06_cfg2_dense.txt gpurun_out/r06_cfg2_dense_graph_timeline.txt
BENCH_ARGS="" bash profiles/pmc_passes.sh r06_cfg2; mv gpurun_out/pmc_r06_cfg2.txt gpurun_out/r06_cfg2_pmc_per_kernel.txt
BENCH_ARGS="--config cfg4" bash profiles/pmc_passes.sh r06_cfg4; mv gpurun_out/pmc_r06_cfg4.txt gpurun_out/r06_cfg4_pmc_per_kernel.txt
BENCH_ARGS="--config cfg5" bash profiles/pmc_passes.sh r06_cfg5; mv gpurun_out/pmc_r06_cfg5.txt gpurun_out/r06_cfg5_pmc_per_kernel.txt
ls -la gpurun_out | grep r06
