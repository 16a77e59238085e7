# kernel timeline inside the replayed HIP graph (GPU box): bash tools/run_timeline2.sh TAG [bench args]  -> gpurun_out/tl_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --kernel-trace -d /tmp/out_$TAG -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --profile-steps 0 "$@" > $R/gpurun_out/tl_$TAG.json 2> /tmp/err_$TAG.txt
python3 $R/tools/graph_timeline.py /tmp/out_$TAG/kt_results.db > $R/gpurun_out/tl_$TAG.txt 2>&1
