# like ab_val.sh with three rounds: bash tools/probe/ab_val3.sh VAR v1 v2 ...
V=$1; shift
for i in 1 2 3; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset $V; else export $V=$v; fi
  bash tools/probe/run_var.sh | sed "s/^/$V=$v  /"
done; done
