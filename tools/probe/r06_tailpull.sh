# quick same-box check of the pulled-tile x16 tails (FfnParams::pull): parity tests, then timings against the one-tile-per-workgroup launch
# (HMVIT_X16_STATIC=1 needs a probe library: HMVIT_ENV is compiled out of the shipped one)
mkdir -p gpurun_out/r06
P=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
timeout 1200 python -m pytest tests/test_hip_fusion.py -x -q -m gpu -k "${1:-split or mixed}" 2>&1 | tail -3
for r in 1 2; do
for v in 0 1; do
  for prec in split mixed; do
  if [ $v = 1 ]; then export HMVIT_X16_STATIC=1; else unset HMVIT_X16_STATIC; fi
  HMVIT_LIB=$P timeout 300 python bench.py --precision $prec --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('static=$v $prec', round(d['ms_per_step'],3), {k: round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
  done
done; done
