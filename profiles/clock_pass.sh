# effective shader clock per kernel (MI355X_MICROARCH.md, DVFS: clock = GRBM_GUI_ACTIVE / kernel wall time): one --pmc pass over the eager step
#   gpurun -- bash profiles/clock_pass.sh TAG   (BENCH_ARGS="--config cfg4" for the other workloads)  -> gpurun_out/clock_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-graph --profile-steps 0 --settle 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/clk_$TAG -o c -- $CMD > /dev/null 2> $R/gpurun_out/clk_$TAG.err
python3 $R/profiles/pmc_summary.py $R/gpurun_out/clk_$TAG/c_results.db > $R/gpurun_out/clock_$TAG.txt
rm -rf $R/gpurun_out/clk_$TAG
