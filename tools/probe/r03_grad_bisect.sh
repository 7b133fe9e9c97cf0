# usage: r03_grad_bisect.sh "<extra CXXFLAGS>" ...   (each variant: rebuild, 6 backward passes, run-to-run spread)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 $v" 2>&1 | grep -E " error" | head
  echo "== $v"; timeout 600 python tests/tools/grad_debug.py det 2>&1 | grep -E "^\[det"
done
