# A/B on one box: x16 chain kernels, one-barrier skew of waves 4-7 (HMVIT_X16_DEPHASE) x LDS fragment prefetch depth (HMVIT_X16_DEPTH).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
run() {  # $1 = label, $2 = extra flags
  make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable $2" > /dev/null 2>&1
  for p in split mixed; do
    python bench.py --precision $p --steps 10 --warmup 2 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', '$p', round(r['ms_per_step'],3), {k:v['ms_total'] for k,v in r['phases'].items()})"
  done
}
run lag0 ""
timeout 900 python -m pytest tests/test_hip_fusion.py -x -q -k "(split or mixed) and (g4 or g5 or full_size or native)" 2>&1 | tail -2
run lag1 "-DHMVIT_X16_REFILL_LAG=1"
timeout 900 python -m pytest tests/test_hip_fusion.py -x -q -k "(split or mixed) and (g4 or g5 or full_size or native)" 2>&1 | tail -2
run lag2 "-DHMVIT_X16_REFILL_LAG=2"
run lag2_d12 "-DHMVIT_X16_REFILL_LAG=2 -DHMVIT_X16_DEPTH=12"
run lag1_inphase "-DHMVIT_X16_REFILL_LAG=1 -DHMVIT_X16_DEPHASE=0"
