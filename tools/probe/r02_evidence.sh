# Round-2 evidence run: the GPU suite, smoke(), the default bench line, then kernel-trace + PMC profiles of the three bench modes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r02/gputest.log
grep -E "FAILED|ERROR|passed|failed|rc=" gpurun_out/r02/gputest.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err; echo "bench rc=$?"
for p in split mixed f16; do
  timeout 900 bash tools/probe/profile_split.sh $p gpurun_out/r02/prof_$p > gpurun_out/r02/prof_$p.log 2>&1; echo "prof $p rc=$?"
done
cut -c1-600 gpurun_out/r02/bench.json
