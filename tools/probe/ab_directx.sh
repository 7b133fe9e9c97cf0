mkdir -p gpurun_out/r05
for r in 1 2 3 4; do
  for v in 0 1; do
    if [ $v = 1 ]; then export HMVIT_NO_DIRECT_X=1; else unset HMVIT_NO_DIRECT_X; fi
    HMVIT_LIB=tools/probe/lib_probe.so python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('no_direct_x=$v', round(d['ms_per_step'],3), {k:round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
  done
done
