cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_width -o kt -- python3 $R/tools/gpu_width_bench.py 32 50 64 > $R/gpurun_out/width.txt 2> $R/gpurun_out/kt_width.err
python3 $R/profiles/summarize.py $R/gpurun_out/kt_width/kt_results.db > $R/gpurun_out/kt_width.txt
rm -rf $R/gpurun_out/kt_width
