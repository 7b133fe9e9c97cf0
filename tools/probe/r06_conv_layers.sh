# per-launch durations of the PointPillar encoder's kernels (f16 / split / f32 runs of tests/tools/encoder_bench.py), grouped by kernel and grid
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06/conv_layers; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- python3 tests/tools/encoder_bench.py > $OUT/run.log 2>&1
cat $OUT/run.log | grep PointPillar
python3 - <<PY
import csv, glob, collections, re
rows = []
for f in glob.glob("$OUT/kt/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = collections.OrderedDict()
for r in rows:
    name = re.sub(r"^void hmvit::", "", r["Kernel_Name"])[:60]
    key = (name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", ""))
    g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in g.items():
    v.sort(); med = v[len(v) // 2]
    if "conv" in k[0] or "pfn" in k[0] or "absmax" in k[0]:
        print(f"{k[0]:60s} grid {k[1]:>8s} n {len(v):4d} median {med:8.1f} us")
PY
rm -rf $OUT/kt
