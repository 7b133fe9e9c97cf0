cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03/conv_prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python3 tests/tools/encoder_bench.py split > $OUT/log.txt 2>&1
grep PointPillar $OUT/log.txt
head -14 $OUT/kt/*kernel_stats.csv | cut -c1-160
python3 tests/tools/train_bench.py cfg2 3 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt2 -o kt -- python3 tests/tools/train_bench.py cfg2 2 > $OUT/log2.txt 2>&1
head -16 $OUT/kt2/*kernel_stats.csv | cut -c1-160
