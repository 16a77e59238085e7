# same-box A/B of alternative library builds (graph_neural_net_amd/_dbg/libfgnn_hip_<name>.so; "main" = the shipped one):
# in-graph kernel timelines + ms/step.  usage (GPU box): bash tools/ab_libs.sh main aux16 ...   [BENCH_ARGS="--input dense ..."]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for NAME in "$@"; do
  TAG=lib_$NAME
  if [ "$NAME" = main ]; then unset FGNN_LIB; else export FGNN_LIB=$R/graph_neural_net_amd/_dbg/libfgnn_hip_$NAME.so; fi
  rocprofv3 --kernel-trace -d /tmp/out_$TAG -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-steps 0 $BENCH_ARGS > $R/gpurun_out/tl_$TAG.json 2> /tmp/err_$TAG.txt
  python3 $R/tools/graph_timeline.py /tmp/out_$TAG/kt_results.db > $R/gpurun_out/tl_$TAG.txt 2>&1
  echo "== $NAME"; head -4 $R/gpurun_out/tl_$TAG.txt | tail -3; tail -1 $R/gpurun_out/tl_$TAG.txt
  python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/tl_$TAG.json').read().strip().split('\n')[-1]); print('ms_per_step', d['ms_per_step'])"
done
