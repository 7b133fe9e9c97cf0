cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1200 python -m pytest tests/test_hip_fusion.py -x -q -k "split" > gpurun_out/r02/split_test.log 2>&1; echo "rc=$?" >> gpurun_out/r02/split_test.log
tail -30 gpurun_out/r02/split_test.log
timeout 600 python -m pytest tests/test_hip_train.py tests/test_hip_fusion.py -q -k "not split" > gpurun_out/r02/rest_test.log 2>&1; echo "rc=$?" >> gpurun_out/r02/rest_test.log
tail -5 gpurun_out/r02/rest_test.log
timeout 300 python bench.py --precision split --steps 10 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/r02/bench_split.json 2> gpurun_out/r02/bench_split.err
tail -3 gpurun_out/r02/bench_split.err
python -c "
import json; r=json.load(open('gpurun_out/r02/bench_split.json')); print(r['value'], r['ms_per_step'], {k:v['ms_total'] for k,v in r['phases'].items()})"
