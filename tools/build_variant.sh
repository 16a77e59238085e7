# one-off debug / ablation build of the library: bash tools/build_variant.sh NAME file.hip[,file2.hip] FLAGS...  -> graph_neural_net_amd/_dbg/libfgnn_hip_NAME.so
# (the listed sources are recompiled with FLAGS, everything else is linked from the regular objects)
set -e
cd "$(dirname "$0")/../graph_neural_net_amd/csrc"
NAME=$1; FILES=$2; shift 2
mkdir -p ../_dbg/$NAME
OBJS=""
for o in *.o; do
  keep=1
  for f in ${FILES//,/ }; do [ "${f%.hip}.o" = "$o" ] && keep=0; done
  [ $keep = 1 ] && OBJS="$OBJS $o"
done
for f in ${FILES//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -Wno-unused-function "$@" -c $f -o ../_dbg/$NAME/${f%.hip}.o
  OBJS="$OBJS ../_dbg/$NAME/${f%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../_dbg/libfgnn_hip_$NAME.so $OBJS
echo built ../_dbg/libfgnn_hip_$NAME.so
