# kernel-trace stats of one cfg2 training step of the fusion (top kernels), optionally for another library build: HMVIT_LIB=... bash tools/probe/train_stats.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=/tmp/tk; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python3 tests/tools/train_bench.py cfg2 1 > $OUT/kt.log 2>&1
cp $OUT/kt/*kernel_stats.csv gpurun_out/r05/tk_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r05/tk_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):4d} calls  {float(r['TotalDurationNs']) / 4e6:7.2f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
