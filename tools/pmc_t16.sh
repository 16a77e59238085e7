# SQ counters of the pair backward (32-pixel and t16) on the GPU box: bash tools/pmc_t16.sh TAG  -> gpurun_out/pmc_t16_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
CMD="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-graph --profile-steps 0 --settle 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/pmc1_$TAG -o p1 -- $CMD > /dev/null 2> /tmp/pmc1_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d /tmp/pmc2_$TAG -o p2 -- $CMD > /dev/null 2> /tmp/pmc2_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 -d /tmp/pmc3_$TAG -o p3 -- $CMD > /dev/null 2> /tmp/pmc3_$TAG.err
python3 $R/profiles/pmc_summary.py /tmp/pmc1_$TAG/p1_results.db /tmp/pmc2_$TAG/p2_results.db /tmp/pmc3_$TAG/p3_results.db > $R/gpurun_out/pmc_t16_$TAG.txt 2> $R/gpurun_out/pmc_t16_$TAG.err
tail -3 /tmp/pmc3_$TAG.err
