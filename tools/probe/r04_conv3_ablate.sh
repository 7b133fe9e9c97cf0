# timing-only ablations of k_conv3r (results are wrong in these builds): which part of a tap costs what
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ "$v" = "base" ]; then unset HMVIT_LIB; else export HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_$v.so; fi
  OUT=gpurun_out/r04/c3_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- python3 tests/tools/encoder_bench.py > $OUT/run.log 2>&1
  python3 - <<PY
import csv, glob, collections, re
rows = []
for f in glob.glob("$OUT/kt/*kernel_trace.csv"): rows += list(csv.DictReader(open(f)))
g = collections.OrderedDict()
for r in rows:
    if "conv" not in r["Kernel_Name"] or "pack" in r["Kernel_Name"] or "true" not in r["Kernel_Name"]: continue
    m = re.search(r"conv3r<(\d+), (\w+)>|conv3rILi(\d+)ELb(\d)", r["Kernel_Name"])
    key = (r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
    g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("$v", {f"{k[0][-22:]}|{k[1]}": round(sorted(v)[len(v)//2], 1) for k, v in g.items()})
PY
  rm -rf $OUT/kt
done
