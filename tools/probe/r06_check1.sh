# round 6 check: the test files the patch kernel / pack_small / bench changes touch, then the default bench line
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_fusion.py tests/test_hip_ops.py -x -q -m gpu 2>&1 | tail -6
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06/bench_check1.json 2> gpurun_out/r06/bench_check1.err; tail -3 gpurun_out/r06/bench_check1.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_check1.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "host", d.get("host_ms_per_forward"))
print("patch", {k: d["patch_attention"][k] for k in ("value", "ms_per_step")}, {k: v["ms_total"] for k, v in d["patch_attention"]["phases"].items() if v["ms_total"] > 0.05})
for k, v in (d.get("roofline_by_phase") or {}).items():
    print(k, round(v["avg_launch_ms"], 3), "mfma", round(v["mfma"]["frac"], 4), "hbm", round(v["hbm"]["frac"], 4), "traffic", v["traffic"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "bound", "frac", "traffic")}, d["roofline"]["traffic_source"]["match"])
print("fast_f16", d["fast_f16"]["value"], "mixed", d["mixed_a16"]["value"], "dense", d["dense_masked_tiles"]["value"], "train", d["train_step"]["ms_per_step"])
print("encoders", d["encoders"], "\nmodel", d["model_e2e"])
PY
