# A/B timing experiments on the split chain kernels (results are WRONG by construction: timing only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # $1 = label, $2 = extra flags
  make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable $2" > /dev/null 2>&1
  python bench.py --precision split --steps 10 --warmup 2 --no-cpu-baseline --no-strict 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],3), {k:v['ms_total'] for k,v in r['phases'].items()})"
}
run stag_0 "-DHMVIT_STAGGER_LN=0 -DHMVIT_STAGGER_TAIL=0"
run stag_5_11 ""
run stag_3_6 "-DHMVIT_STAGGER_LN=3 -DHMVIT_STAGGER_TAIL=6"
run stag_8_16 "-DHMVIT_STAGGER_LN=8 -DHMVIT_STAGGER_TAIL=16"
