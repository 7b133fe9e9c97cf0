# one-off: parity of the fusion forward + the default bench line with whatever libhmvit.so travelled with the snapshot
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_fusion.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v['ms_total'] for k, v in d['phases'].items()})"
