# training kernels after a change: operator + gradient parity tests, reproducibility of the backward, then the cfg2 train-step figure
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_fusion.py -m gpu -x -q -k "linear16 or generic_window" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
timeout 2400 python -m pytest tests/test_hip_train.py tests/test_hip_trainer.py -m gpu -x -q -s > gpurun_out/r03/train_check.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r03/train_check.log | tail -5
for i in 1 2; do timeout 600 python tools/probe/bwd_repro.py 2>&1 | grep "run to run" | cut -c1-120; done
timeout 600 python tests/tools/train_bench.py cfg2 3 2>&1 | tail -1
