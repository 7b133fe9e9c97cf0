# headline line only (no side figures, no CPU baseline), phases printed: python bench.py --no-strict --no-cpu-baseline
#   bash tools/probe/quick_bench.sh [lib.so ...]    (default: the shipped library)
mkdir -p gpurun_out/r05
for L in "${@:-hm-vit_amd/libhmvit.so}"; do
  HMVIT_LIB=$L python bench.py --no-strict --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$L', round(d['value'],2),'scenes/s', round(d['ms_per_step'],3),'ms', {k:round(v['ms_total'],3) for k,v in d['phases'].items() if v['ms_total']>0.05})"
done
