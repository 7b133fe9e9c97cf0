# one-off: run the tests given as arguments (pytest node ids / -k expressions), tail of the output
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest "$@" -m gpu -x -q -s 2>&1 | tail -40
