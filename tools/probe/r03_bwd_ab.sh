cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "-DHMVIT_DBG_BWD_OCC2"; do
  make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 $v" 2>&1 | grep -E " error" | head
  echo "== $v"
  rm -rf gpurun_out/r03/bwd_ab; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r03/bwd_ab -o kt -- python3 tests/tools/train_bench.py cfg2 2 > /dev/null 2>&1
  grep -E "k_attention_bwd|k_gemm_split" gpurun_out/r03/bwd_ab/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
  python3 tests/tools/train_bench.py cfg2 3 2>&1 | tail -1
done
