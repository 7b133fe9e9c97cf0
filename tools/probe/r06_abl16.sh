for v in nodma nomath noblend none; do
  HMVIT_PATCH_ATTENTION=2 HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_q_$v.so timeout 300 python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
done
