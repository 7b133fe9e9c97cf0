# per-launch duration and HBM fetch of the split attention kernel (4 launches per forward: local, grid, local (pruned prev), grid (ego 0))
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06/attn_prof_${1:-a}
mkdir -p $OUT
B="python3 bench.py --precision split --steps 4 --warmup 1 --no-cpu-baseline --no-strict"
rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d $OUT/tc -o p -- $B > $OUT/tc.log 2>&1
python3 - <<PY
import csv, glob, collections
def rows(pat):
    for f in glob.glob(pat):
        yield from csv.DictReader(open(f))
d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows("$OUT/kt/*kernel_trace.csv") if "k_attention_p" in r["Kernel_Name"]]
d.sort()
per = collections.defaultdict(list)
for i, (_, us) in enumerate(d):
    per[i % 4].append(us)
print("duration us per stage (median):", {k: sorted(v)[len(v) // 2] for k, v in per.items()})
for name, pat in (("FETCH_SIZE", "$OUT/fetch/*counter_collection.csv"), ("TCC", "$OUT/tc/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rs = [r for r in rows(pat) if "k_attention_p" in r["Kernel_Name"]]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    ids = sorted({int(r["Dispatch_Id"]) for r in rs})
    pos = {d: i % 4 for i, d in enumerate(ids)}
    for r in rs:
        acc[r["Counter_Name"]][pos[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
    for cn, st in acc.items():
        print(cn, {k: round(sorted(v)[len(v) // 2] * (2 * 1024 / 1e9 if cn == "FETCH_SIZE" else 1), 3) for k, v in sorted(st.items())}, "(FETCH: GB, x2 corrected)" if cn == "FETCH_SIZE" else "")
PY
