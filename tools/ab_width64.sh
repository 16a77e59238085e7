# same-box A/B of library builds on the 64-feature step: bash tools/ab_width64.sh main NAME ...  (tools/build_variant.sh NAME mlp64.hip -D...)
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = main ]; then unset FGNN_LIB; else export FGNN_LIB=$GRAFT_REPO_ROOT/graph_neural_net_amd/_dbg/libfgnn_hip_$v.so; fi
  echo "== $v"; timeout 200 python tools/gpu_width_bench.py 32 50 64
done
