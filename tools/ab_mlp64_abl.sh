export PYTHONPATH=$GRAFT_REPO_ROOT
for v in main m64a1 m64a2 m64a3; do
  if [ $v = main ]; then unset FGNN_LIB; else export FGNN_LIB=$GRAFT_REPO_ROOT/graph_neural_net_amd/_dbg/libfgnn_hip_$v.so; fi
  echo "== $v"; timeout 200 python tools/gpu_mlp64_probe.py | grep -E "cin  64|cin 128"
done
