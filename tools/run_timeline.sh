# kernel timeline inside the replayed HIP graph (GPU box): bash tools/run_timeline.sh [f32|x3 ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
for M in ${@:-x3}; do
  export FGNN_MFMA=$M
  rocprofv3 --kernel-trace -d /tmp/out_$M -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-steps 0 > $R/gpurun_out/r3/tl_bench_$M.json 2> /tmp/err_$M.txt
  python3 $R/tools/graph_timeline.py /tmp/out_$M/kt_results.db > $R/gpurun_out/r3/timeline_$M.txt 2>&1
done
