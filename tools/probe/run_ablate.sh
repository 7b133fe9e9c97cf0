for lib in tools/probe/lib_*.so; do
HMVIT_LIB=$PWD/$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/ab.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/ab.log") if x.startswith("{")]
d=json.loads(l[-1]); print("$lib", {k: round(v["ms_total"],3) for k,v in d["phases"].items() if k in ("qkv_gemm","ffn2")})
PY
done
