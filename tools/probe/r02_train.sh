cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1200 python -m pytest tests/test_hip_train.py -x -q > gpurun_out/r02/train_test.log 2>&1; echo "rc=$?" >> gpurun_out/r02/train_test.log
tail -40 gpurun_out/r02/train_test.log
timeout 900 python -m pytest tests/test_hip_fusion.py tests/test_hip_encoder.py -x -q > gpurun_out/r02/fusion_test.log 2>&1; echo "rc=$?" >> gpurun_out/r02/fusion_test.log
tail -5 gpurun_out/r02/fusion_test.log
timeout 300 python bench.py --precision f16 --steps 20 --warmup 3 --no-cpu-baseline --no-strict > gpurun_out/r02/bench_f16.json 2> gpurun_out/r02/bench_f16.err
python -c "
import json; r=json.load(open('gpurun_out/r02/bench_f16.json')); print(r['value'], r['ms_per_step'], {k:v['ms_total'] for k,v in r['phases'].items()})"
