# same-box comparison of values of one environment variable: bash tools/probe/ab_val.sh VAR v1 v2 ... ("-" = unset)
V=$1; shift
for i in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset $V; else export $V=$v; fi
  bash tools/probe/run_var.sh | sed "s/^/$V=$v  /"
done; done
