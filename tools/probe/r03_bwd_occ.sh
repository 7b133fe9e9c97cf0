# the attention backward at two workgroups per CU: run-to-run reproducibility with probe defines
# usage: r03_bwd_occ.sh "<defines>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for fl in "$@"; do
make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-variable $fl" 2>&1 | grep -E " error" -A3 | head
echo "== flags: $fl"
for i in 1 2 3; do timeout 900 python tools/probe/bwd_repro.py 2>&1 | grep -E "run to run" | cut -c1-140; done; done
