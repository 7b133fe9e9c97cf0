# quick same-box check of the patch kernel: a few parity tests, trace of workgroup 0, then shipped / gather timings (probe library)
mkdir -p gpurun_out/r06
P=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
timeout 600 python -m pytest tests/test_hip_fusion.py -x -q -m gpu -k "${2:-full_size or native_window8 or two_five}" 2>&1 | tail -3
HMVIT_PATCH_ATTENTION=1 HMVIT_LIB=$P python tests/tools/patch_trace.py 2>/dev/null | head -${1:-30}
for r in 1 2; do
for v in patch gather; do
  if [ $v = patch ]; then export HMVIT_PATCH_ATTENTION=1; else unset HMVIT_PATCH_ATTENTION; fi
  if [ $v = gather ]; then export HMVIT_NO_PATCH=1; else unset HMVIT_NO_PATCH; fi
  HMVIT_LIB=$P timeout 300 python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
done; done
