# same-box A/B of an opt-in environment switch: bash tools/probe/ab_env_on.sh VAR
for i in 1 2 3; do
  unset $1; bash tools/probe/run_var.sh | sed 's/^/off  /'
  export $1=1; bash tools/probe/run_var.sh | sed "s/^/on   /"
done
