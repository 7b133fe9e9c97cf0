# same-box A/B of probe libraries: bash tools/probe/r06_ab_libs.sh <name> <name> ...   (tools/probe/lib_<name>.so, two rounds, split + mixed)
for r in 1 2 3; do
for v in "$@"; do
  for prec in split; do
  HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_$v.so timeout 300 python bench.py --precision $prec --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $prec', round(d['ms_per_step'],3), {k: round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
  done
done; done
