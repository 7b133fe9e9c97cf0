# same-box A/B: default library against tools/probe/libhmvit_base.so (HMVIT_LIB override), alternating
for i in 1 2 3; do
  unset HMVIT_LIB; bash tools/probe/run_var.sh | sed 's/^/new  /'
  export HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/libhmvit_base.so; bash tools/probe/run_var.sh | sed 's/^/base /'
done
