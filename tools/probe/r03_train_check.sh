# training kernels after a change: operator + gradient parity tests, then the cfg2 train-step figure
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -s -k "linear16" 2>&1 | grep -E "linear16 M|passed|failed|Error|assert" | head -20
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_trainer.py -m gpu -x -q -s > gpurun_out/r03/train_check.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r03/train_check.log | tail -5
timeout 600 python tests/tools/train_bench.py cfg2 3 2>&1 | tail -1
