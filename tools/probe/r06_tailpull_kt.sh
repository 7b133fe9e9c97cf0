# per-launch durations of the attention and tail kernels with pulled and one-tile-per-workgroup tails (probe library: HMVIT_X16_STATIC)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
B="python3 bench.py --precision split --steps 6 --warmup 2 --no-cpu-baseline --no-strict"
for v in 0 1 0 1; do
  OUT=gpurun_out/r06/tailpull_kt_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = 1 ]; then export HMVIT_X16_STATIC=1; else unset HMVIT_X16_STATIC; fi
  rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- $B > $OUT/kt.log 2>&1
  python3 - <<PY
import csv, glob, collections
rows = [r for f in glob.glob("$OUT/kt/*kernel_trace.csv") for r in csv.DictReader(open(f))]
for kn, n in (("k_attention_p", 4), ("k_out_ffn_qkv16", 3), ("k_out_ffn_head16", 1), ("k_ln_qkv16", 1), ("k_tile_vis", 4)):
    d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if kn in r["Kernel_Name"])
    per = collections.defaultdict(list)
    for i, (_, us) in enumerate(d): per[i % n].append(us)
    print("static=$v", kn, {k: round(sorted(x)[len(x) // 2], 1) for k, x in per.items()})
PY
done
