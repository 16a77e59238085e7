# kernel-trace pass only (per-kernel average durations): bash profiles/kt_pass.sh TAG   with BENCH_ARGS="--config cfg5"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --profile-steps 0 --settle 0 --no-extra-configs ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$TAG -o kt -- $CMD > /dev/null 2> $R/gpurun_out/kt_$TAG.err
python3 $R/profiles/summarize.py $R/gpurun_out/kt_$TAG/kt_results.db > $R/gpurun_out/kt_$TAG.txt
rm -rf $R/gpurun_out/kt_$TAG
