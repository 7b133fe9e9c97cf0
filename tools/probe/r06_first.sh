# round 6, first GPU contact of k_attention_patch: fusion parity tests, then a same-box A/B of the patch kernel against the gather kernel
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_fusion.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/first_tests.txt
cat gpurun_out/r06/first_tests.txt
P=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
for r in 1 2; do
  HMVIT_LIB=$P timeout 300 python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | sed 's/^/patch  /' >> gpurun_out/r06/first_ab.txt
  HMVIT_LIB=$P HMVIT_NO_PATCH=1 timeout 300 python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | sed 's/^/gather /' >> gpurun_out/r06/first_ab.txt
done
python - <<'PY'
import json
for line in open("gpurun_out/r06/first_ab.txt"):
    tag, js = line.split(None, 1)
    try: d = json.loads(js)
    except Exception as e: print(tag, "FAILED", js[:200]); continue
    print(tag, d["value"], d["ms_per_step"], {k: round(v["ms_total"], 3) for k, v in d["phases"].items() if v["ms_total"] > 0.05})
PY
