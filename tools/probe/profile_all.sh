# Round profile: kernel-trace stats, then separate PMC passes (never combined with other trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict"
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof/kt -o kt -- $B > gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/prof/fetch -o p -- $B > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d gpurun_out/prof/write -o p -- $B > gpurun_out/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD -f csv -d gpurun_out/prof/sq1 -o p -- $B > gpurun_out/prof_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS -f csv -d gpurun_out/prof/sq2 -o p -- $B > gpurun_out/prof_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -f csv -d gpurun_out/prof/tc -o p -- $B > gpurun_out/prof_tc.log 2>&1
ls -R gpurun_out/prof | head -40
