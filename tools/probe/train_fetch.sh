# HBM bytes fetched per kernel over one cfg2 training step (FETCH_SIZE, x2 wide-read correction applied in the print)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=/tmp/tf; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/p -o p -- python3 tests/tools/train_bench.py cfg2 1 > $OUT/log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tf/p/*counter_collection.csv") + glob.glob("/tmp/tf/p/*/*counter_collection.csv")
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k, v in tot.most_common(8):
    print(f"{k:42s} {n[k]:5d} launches  {v * 64 * 2 / 1e9 / 4:8.2f} GB fetched per step (4 steps in the trace)")
PY
