# same-box A/B of the 16-pixel-tile kernels: in-graph kernel timelines + ms/step for FGNN_T16 = 0 and the given selections
# usage (GPU box): bash tools/ab_t16.sh "pair" "pair,bwd" ...   -> gpurun_out/tl_t16_<sel>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SEL in 0 "$@"; do
  TAG=t16_$(echo $SEL | tr ',' '_')
  export FGNN_T16=$SEL
  rocprofv3 --kernel-trace -d /tmp/out_$TAG -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-steps 0 $BENCH_ARGS > $R/gpurun_out/tl_$TAG.json 2> /tmp/err_$TAG.txt
  python3 $R/tools/graph_timeline.py /tmp/out_$TAG/kt_results.db > $R/gpurun_out/tl_$TAG.txt 2>&1
  echo "== FGNN_T16=$SEL"; head -8 $R/gpurun_out/tl_$TAG.txt; tail -1 $R/gpurun_out/tl_$TAG.txt
  python3 -c "import json,sys; d=json.loads(open('$R/gpurun_out/tl_$TAG.json').read().strip().split('\n')[-1]); print('ms_per_step', d['ms_per_step'], 'dense', d.get('dense_input',{}).get('ms_per_step'))"
done
