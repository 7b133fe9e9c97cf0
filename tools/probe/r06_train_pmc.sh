# kernel trace + PMC passes (separate runs, never combined with other trace domains) of the cfg2 training step of the fusion
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06/train_pmc
rm -rf $OUT; mkdir -p $OUT
B="python3 tests/tools/train_bench.py cfg2 1"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o p -- $B > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD -f csv -d $OUT/sq1 -o p -- $B > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS -f csv -d $OUT/sq2 -o p -- $B > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -f csv -d $OUT/tc -o p -- $B > $OUT/tc.log 2>&1
python3 tools/pmc_by_kernel.py $OUT gpurun_out/r06/train_pmc.txt 9
cp $OUT/kt/*kernel_stats.csv gpurun_out/r06/train_kernel_stats.csv
rm -rf $OUT
