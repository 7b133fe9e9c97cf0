cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 3000 python -m pytest tests -m gpu -q --durations=40 > gpurun_out/r03/durations.log 2>&1
grep -E "passed|failed" gpurun_out/r03/durations.log | tail -2
grep -E "^[0-9.]+s (call|setup)" gpurun_out/r03/durations.log | head -40
