cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q "$@" > gpurun_out/r02/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r02/gputest.log
grep -E "FAILED|ERROR|passed|failed|rc=" gpurun_out/r02/gputest.log | head -40
