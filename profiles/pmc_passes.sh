cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
CMD="python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph --profile-steps 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/pmc1 -o p1 -- $CMD > /dev/null 2> $R/gpurun_out/pmc1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc2 -o p2 -- $CMD > /dev/null 2> $R/gpurun_out/pmc2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o p3 -- $CMD > /dev/null 2> $R/gpurun_out/pmc3.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc4 -o p4 -- $CMD > /dev/null 2> $R/gpurun_out/pmc4.err
ls -la $R/gpurun_out/pmc*/ | head -30
tail -3 $R/gpurun_out/pmc1.err $R/gpurun_out/pmc2.err
