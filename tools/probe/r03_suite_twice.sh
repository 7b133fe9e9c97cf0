# the whole GPU suite twice on one box: flakiness check of the statistically bounded tests before the round ends
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
done
