cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 PROBE=1 2>&1 | grep -E "error" | head
timeout 600 python tools/probe/bwd_trace.py 2>&1 | tail -8
