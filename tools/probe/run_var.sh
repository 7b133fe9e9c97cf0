for v in ${VARS:-none}; do
if [ "$v" = "none" ]; then unset HMVIT_ATTN_DEBUG; else export HMVIT_ATTN_DEBUG=$v; fi
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/b$v.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/b$v.log") if x.startswith("{")]
d=json.loads(l[-1]); print("$v", round(d["value"],2), {k: round(x["ms_total"],3) for k,x in d["phases"].items()})
PY
done
