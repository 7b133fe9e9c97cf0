# round 3, first GPU call: range stress tests on the unchanged kernels + a baseline bench of the split mode
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_range.py -q -s 2>&1 | grep -E "range\[|passed|failed|Error" > gpurun_out/r03/range_before.txt
tail -30 gpurun_out/r03/range_before.txt
timeout 300 python bench.py --precision split --steps 10 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/r03/bench_split_before.json 2> gpurun_out/r03/bench_split_before.err
python -c "
import json,sys; r=json.load(open('gpurun_out/r03/bench_split_before.json')); print(r['value'], r['ms_per_step'], {k:v['ms_total'] for k,v in r['phases'].items()})"
