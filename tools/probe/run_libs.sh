# same-box timing of library variants: bash tools/probe/run_libs.sh [bench flags --] name1 name2 ...   ("base" = the tree's library)
mkdir -p gpurun_out/r04
FLAGS="--precision split --steps 10 --warmup 3 --no-cpu-baseline --no-strict"
for v in "$@"; do
  if [ "$v" = "base" ]; then unset HMVIT_LIB; else export HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_$v.so; fi
  python bench.py $FLAGS > gpurun_out/r04/lib_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r04/lib_$v.log") if x.startswith("{")]
if not l: print("$v FAILED"); raise SystemExit
d=json.loads(l[-1]); print("$v", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k: round(x["ms_total"],3) for k,x in d["phases"].items()})
PY
done
