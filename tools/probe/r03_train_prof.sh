# kernel trace of the cfg2 training step of the fusion (tests/tools/train_bench.py): per-kernel totals -> gpurun_out/r03/train_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03/train_kt
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r03/train_kt -o tr -- python3 tests/tools/train_bench.py cfg2 3 > gpurun_out/r03/train_prof.log 2>&1
tail -2 gpurun_out/r03/train_prof.log
f=$(find gpurun_out/r03/train_kt -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r03/train_kernel_stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r03/train_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} total {float(r["TotalDurationNs"])/1e6:9.2f} ms  avg {float(r["AverageNs"])/1e3:9.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total", tot / 1e6, "ms over 7 steps (1 warm-up + 3 + 3)")
PY
rm -rf gpurun_out/r03/train_kt
