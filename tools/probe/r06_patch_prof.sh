# round 6: where k_attention_patch's time goes - role ablations (timing-only libraries built by tools/probe/build_var.sh p_<name>),
# per-launch durations and SQ / LDS / TD counters of the shipped kernel (separate PMC passes)
#   bash tools/probe/r06_patch_prof.sh [tag] [variants...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-a}; shift
OUT=gpurun_out/r06/patch_prof_$TAG; rm -rf $OUT; mkdir -p $OUT
VARS=${@:-nodma noblend nomath notables mathonly dmaonly none}
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | sed "s/^/$n /" >> $OUT/ab.txt
}
run shipped HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
run gather HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so HMVIT_NO_PATCH=1
for v in $VARS; do run $v HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_p_$v.so; done
run shipped HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
python3 - <<PY
import json
for line in open("$OUT/ab.txt"):
    tag, js = line.split(None, 1)
    try: d = json.loads(js)
    except Exception: print(tag, "FAILED"); continue
    print(f"{tag:10s} {d['ms_per_step']:.3f} ms", {k: round(v["ms_total"], 3) for k, v in d["phases"].items() if v["ms_total"] > 0.05})
PY
B="python3 bench.py --precision split --steps 4 --warmup 1 --no-cpu-baseline --no-strict"
timeout -s KILL 200 rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD -f csv -d $OUT/sq1 -o p -- $B > $OUT/sq1.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS -f csv -d $OUT/sq2 -o p -- $B > $OUT/sq2.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc TD_TD_BUSY_sum TD_TC_STALL_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum -f csv -d $OUT/td -o p -- $B > $OUT/td.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
python3 - <<PY
import csv, glob, collections
def rows(pat):
    for f in glob.glob(pat):
        yield from csv.DictReader(open(f))
for kn in ("k_attention_patch", "k_attention_pcs2"):
    d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows("$OUT/kt/*kernel_trace.csv") if kn in r["Kernel_Name"])
    per = collections.defaultdict(list)
    for i, (_, us) in enumerate(d): per[i % 2].append(us)
    print(kn, "duration us per launch of a forward (median):", {k: sorted(v)[len(v) // 2] for k, v in per.items()})
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for pat in ("sq1", "sq2", "td", "fetch"):
        rs = [r for r in rows(f"$OUT/{pat}/*counter_collection.csv") if kn in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rs})
        pos = {d_: i % 2 for i, d_ in enumerate(ids)}
        for r in rs: acc[r["Counter_Name"]][pos[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
    for cn, st in sorted(acc.items()):
        print(f"  {cn:32s}", {k: f"{sorted(v)[len(v) // 2]:.3e}" for k, v in sorted(st.items())})
PY
