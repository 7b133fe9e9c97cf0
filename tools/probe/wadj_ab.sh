# k_warp_adjoint with the fused bias sums: pixels per workgroup (HMVIT_WADJ_RUN x 4) against the launch time, one cfg2 training step each.
# Build the variants first (in the repo, before gpurun):
#   for r in 1 2 4 8; do bash tools/probe/build_var.sh wadj_run$r "-DHMVIT_WADJ_RUN=$r" train.hip; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for L in tools/probe/lib_wadj_run1.so tools/probe/lib_wadj_run2.so tools/probe/lib_wadj_run4.so tools/probe/lib_wadj_run8.so; do
  OUT=/tmp/tk; rm -rf $OUT; mkdir -p $OUT
  HMVIT_LIB=$L timeout -s KILL 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python3 tests/tools/train_bench.py cfg2 1 > $OUT/kt.log 2>&1
  echo "$L: $(grep warp_adjoint $OUT/kt/*kernel_stats.csv | cut -d, -f2-4)"
done
