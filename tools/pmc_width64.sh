# PMC passes (SQ counters, HBM bytes) of the 64-feature model's step (tools/gpu_width_bench.py 32 50 64): -> gpurun_out/pmc_width64.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
CMD="python3 $R/tools/gpu_width_bench.py 32 50 64"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/pmc1_w -o p1 -- $CMD > /dev/null 2> /tmp/pmc1_w.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 -d /tmp/pmc2_w -o p2 -- $CMD > /dev/null 2> /tmp/pmc2_w.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc3_w -o p3 -- $CMD > /dev/null 2> /tmp/pmc3_w.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc4_w -o p4 -- $CMD > /dev/null 2> /tmp/pmc4_w.err
python3 $R/profiles/pmc_summary.py /tmp/pmc1_w/p1_results.db /tmp/pmc2_w/p2_results.db /tmp/pmc3_w/p3_results.db /tmp/pmc4_w/p4_results.db > $R/gpurun_out/pmc_width64.txt 2> $R/gpurun_out/pmc_width64.err
tail -2 /tmp/pmc2_w.err
