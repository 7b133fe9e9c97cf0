# is the headline loop power-limited?  shader clock / power sampled while bench.py loops (rocm-smi as an ordinary user)
mkdir -p gpurun_out/r06
(python bench.py --precision ${1:-split} --steps 5000 --warmup 20 --no-strict --no-cpu-baseline > gpurun_out/r06/clk_bench.log 2>&1 &)
sleep 30
for i in $(seq 1 12); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' | sed 's/  */ /g'; echo; sleep 1; done
wait
tail -1 gpurun_out/r06/clk_bench.log | cut -c1-200
