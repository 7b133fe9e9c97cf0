# round 2, first GPU call: parity suite, then the f32 / f16 bench lines and the f32 kernel profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/gputest.log
tail -5 gpurun_out/r02/gputest.log
timeout 600 python bench.py --precision f32 --steps 4 --warmup 1 > gpurun_out/r02/bench_f32.json 2> gpurun_out/r02/bench_f32.err
tail -c 3000 gpurun_out/r02/bench_f32.json
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r02/kt_f32 -o kt -- python3 bench.py --precision f32 --steps 3 --warmup 1 --no-cpu-baseline --no-strict > gpurun_out/r02/kt_f32.log 2>&1
head -30 gpurun_out/r02/kt_f32/*kernel_stats.csv
