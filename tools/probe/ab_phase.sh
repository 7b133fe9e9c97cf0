# same-box A/B of library builds on the headline forward: R interleaved rounds of `bench.py --no-strict --no-cpu-baseline` per library,
# then mean / min / max of every phase per library (run-to-run spread on one box is ~1.5 %, between boxes ~3 %)
#   bash tools/probe/ab_phase.sh R lib_a.so lib_b.so ...
R=$1; shift
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_phase_raw.txt; : > $OUT
for r in $(seq 1 $R); do
  for L in "$@"; do
    HMVIT_LIB=$L python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | sed "s|^|$L\t|" >> $OUT
  done
done
python - <<'PY'
import json, collections
rows = collections.defaultdict(list)
for line in open("gpurun_out/r05/ab_phase_raw.txt"):
    lib, js = line.rstrip("\n").split("\t", 1)
    d = json.loads(js)
    rows[lib].append({"ms_per_step": d["ms_per_step"], **{k: v["ms_total"] for k, v in d["phases"].items() if v["ms_total"] > 0.05}})
for lib, rs in rows.items():
    keys = rs[0].keys()
    print(lib, len(rs), "runs:", "  ".join(f"{k} {sum(r[k] for r in rs)/len(rs):.3f} [{min(r[k] for r in rs):.3f}, {max(r[k] for r in rs):.3f}]" for k in keys))
PY
