# A/B of attention schedule switches on a probe build (environment switches live only there): per-stage duration and fetch
# usage: r03_attn_ab.sh "VAR=val VAR2=val" "VAR=val" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 PROBE=1 2>&1 | grep -E "error" | head
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== $v"
  env $v bash tools/probe/r03_attn_prof.sh ab$i 2>&1 | tail -4
done
