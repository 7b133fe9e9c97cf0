# same-box A/B of an environment switch: bash tools/probe/ab_env.sh VAR
for i in 1 2 3; do
  unset $1; bash tools/probe/run_var.sh | sed 's/^/on   /'
  export $1=1; bash tools/probe/run_var.sh | sed "s/^/off  /"
done
