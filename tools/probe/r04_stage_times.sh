# per-launch duration of the split attention kernel for library variants: bash tools/probe/r04_stage_times.sh base prev ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = "base" ]; then unset HMVIT_LIB; else export HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_$v.so; fi
  OUT=gpurun_out/r04/st_$v; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- python3 bench.py --precision split --steps 6 --warmup 2 --no-cpu-baseline --no-strict > $OUT/kt.log 2>&1
  python3 - <<PY
import csv, glob, collections
d = []
for f in glob.glob("$OUT/kt/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_attention_pcs" in r["Kernel_Name"]: d.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
d.sort()
per = collections.defaultdict(list)
for i, (_, us) in enumerate(d): per[i % 4].append(us)
m = {k: round(sorted(v)[len(v) // 2]) for k, v in per.items()}
print("$v", "attention us per launch (median of", len(d) // 4, "):", m, "sum", sum(m.values()))
PY
  rm -rf $OUT/kt
done
