# round 3: range stress + parity suites + split bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_range.py -q -s 2>&1 | grep -E "range\[|passed|failed|Error|error" > gpurun_out/r03/range_after.txt
grep -v "range\[split:\|range\[f32:\|range\[decoder:f32\|range\[pointpillar:f32" gpurun_out/r03/range_after.txt | tail -40
if [ "$1" != "quick" ]; then timeout 2400 python -m pytest tests/test_hip_fusion.py tests/test_hip_ops.py tests/test_hip_encoder.py tests/test_hip_model.py -q -x 2>&1 | tail -6; fi
for i in 1 2; do
timeout 300 python bench.py --precision split --steps 10 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/r03/bench_split_scaled.json 2> gpurun_out/r03/bench_split_scaled.err
python -c "
import json,sys; r=json.load(open('gpurun_out/r03/bench_split_scaled.json')); print(r['value'], r['ms_per_step'], {k:v['ms_total'] for k,v in r['phases'].items()})"
done
