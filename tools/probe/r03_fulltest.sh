cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 3300 python -m pytest tests -q -x -m gpu -s 2>&1 | grep -vE "^range\[|^$" | tail -40 > gpurun_out/r03/fulltest.txt
tail -25 gpurun_out/r03/fulltest.txt
