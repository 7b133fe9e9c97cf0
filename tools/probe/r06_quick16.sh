# quick same-box check of k_attention_patch16 (HMVIT_PATCH_ATTENTION=2): parity tests with it forced on, then timings against the other two kernels
mkdir -p gpurun_out/r06
P=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so
HMVIT_PATCH_ATTENTION=2 timeout 900 python -m pytest tests/test_hip_fusion.py -x -q -m gpu -k "${1:-full_size_goldens and split or native_window8 and split or two_five and split}" 2>&1 | tail -3
for r in 1 2; do
for v in 2 1 0; do
  HMVIT_PATCH_ATTENTION=$v HMVIT_LIB=$P timeout 300 python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('patch_attention=$v', round(d['ms_per_step'],3), {k: round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
done; done
