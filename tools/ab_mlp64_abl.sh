# same-box A/B of mlp64 builds (tools/build_variant.sh NAME mlp64.hip -D...): bash tools/ab_mlp64_abl.sh main nwf12 ...
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = main ]; then unset FGNN_LIB; else export FGNN_LIB=$GRAFT_REPO_ROOT/graph_neural_net_amd/_dbg/libfgnn_hip_$v.so; fi
  echo "== $v"; timeout 200 python tools/gpu_mlp64_probe.py | grep -E "G 64"
done
