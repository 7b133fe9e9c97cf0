# where does k_attention_bwd spend a launch?  Probe build, cfg2 train step with parts of the kernel switched off (results are wrong by design)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 PROBE=1 2>&1 | grep -E "error" | head
for v in 0 1 2 3; do
  echo "== HMVIT_BWD_PROBE=$v"
  HMVIT_BWD_PROBE=$v timeout 600 python tests/tools/train_bench.py cfg2 2 2>&1 | tail -1
done
