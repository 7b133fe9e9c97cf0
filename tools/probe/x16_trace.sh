cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C hm-vit_amd/csrc clean > /dev/null; make -C hm-vit_amd/csrc -j16 PROBE=1 > /dev/null 2>&1
python tools/probe/x16_trace.py split
