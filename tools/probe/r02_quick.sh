cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
PREC=${1:-split}
timeout 600 python -m pytest tests/test_hip_fusion.py -x -q -k "$PREC and (g4 or g5 or full_size or native)" 2>&1 | tail -3
timeout 300 python bench.py --precision $PREC --steps 10 --warmup 2 --no-cpu-baseline --no-strict > gpurun_out/r02/bench_$PREC.json 2> gpurun_out/r02/bench_$PREC.err
python -c "
import json,sys; r=json.load(open('gpurun_out/r02/bench_$PREC.json')); print(r['value'], r['ms_per_step'], {k:v['ms_total'] for k,v in r['phases'].items()})"
