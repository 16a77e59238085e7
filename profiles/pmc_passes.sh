# rocprofv3 passes used for the profiles/ summaries (run on the GPU box via gpurun).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-extra-configs --profile-steps 0 --settle 0 ${BENCH_ARGS:-}"   # BENCH_ARGS="--config cfg4" for the bf16 N=200 workload
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$TAG -o kt -- $CMD > /dev/null 2> $R/gpurun_out/kt_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/pmc1_$TAG -o p1 -- $CMD > /dev/null 2> $R/gpurun_out/pmc1_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc2_$TAG -o p2 -- $CMD > /dev/null 2> $R/gpurun_out/pmc2_$TAG.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3_$TAG -o p3 -- $CMD > /dev/null 2> $R/gpurun_out/pmc3_$TAG.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc4_$TAG -o p4 -- $CMD > /dev/null 2> $R/gpurun_out/pmc4_$TAG.err
python3 $R/profiles/summarize.py $R/gpurun_out/kt_$TAG/kt_results.db > $R/gpurun_out/kt_$TAG.txt
python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc1_$TAG/p1_results.db $R/gpurun_out/pmc2_$TAG/p2_results.db $R/gpurun_out/pmc3_$TAG/p3_results.db $R/gpurun_out/pmc4_$TAG/p4_results.db > $R/gpurun_out/pmc_$TAG.txt
rm -rf $R/gpurun_out/pmc1_$TAG $R/gpurun_out/pmc2_$TAG $R/gpurun_out/pmc3_$TAG $R/gpurun_out/pmc4_$TAG $R/gpurun_out/kt_$TAG
