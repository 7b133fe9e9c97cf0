mkdir -p gpurun_out/r04
export HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
for m in dyn static dyn static; do
  if [ $m = static ]; then export HMVIT_PCS_STATIC=1; else unset HMVIT_PCS_STATIC; fi
  python bench.py --precision split --steps 10 --warmup 3 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$m', round(d['value'],2), {k: round(x['ms_total'],3) for k,x in d['phases'].items()})"
done
unset HMVIT_PCS_STATIC; echo dyn; python tools/probe/r04_attn_balance.py 2>&1 | tail -2
export HMVIT_PCS_STATIC=1; echo static; python tools/probe/r04_attn_balance.py 2>&1 | tail -2
