# kernels of the CVT camera encoder (split), per forward: count, total and median duration, sorted by total
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06/cvt_layers; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- python3 tests/tools/camera_bench.py ${1:-split} > $OUT/run.log 2>&1
grep Cvt $OUT/run.log
python3 - <<PY
import csv, glob, collections, re
rows = []
for f in glob.glob("$OUT/kt/*kernel_trace.csv"): rows += list(csv.DictReader(open(f)))
g = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("hmvit::", "")[:46]
    key = (name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
    g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
nf = 4.0   # forwards in the run (1 warm-up + 3 timed)
tot = sum(sum(v) for v in g.values()) / nf
print(f"GPU busy per forward: {tot:.0f} us")
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:28]:
    print(f"{k[0]:46s} grid {k[1]:>9s} n/fwd {len(v)/nf:5.1f} total/fwd {sum(v)/nf:8.1f} us  median {sorted(v)[len(v)//2]:8.1f}")
PY
rm -rf $OUT/kt
