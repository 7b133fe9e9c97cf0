# Round-4 evidence, second half (after the last kernel change): affected GPU tests, smoke(), kernel-trace + PMC profiles of the three
# bench modes, the default bench line (roofline.traffic from the profiles of the same sources), encoder / model figures.
# Every artefact is copied into gpurun_out/r04/profiles/ as soon as it exists (a call that is cut off keeps what it has).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r04/profiles; mkdir -p $P
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r04/gputest_final.log 2>&1; echo "rc=$?" >> gpurun_out/r04/gputest_final.log
grep -E "FAILED|ERROR|passed|failed|rc=" gpurun_out/r04/gputest_final.log | head -20
grep -E "^range\[" gpurun_out/r04/gputest_final.log > $P/r04_range.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.log 2>&1; echo "smoke rc=$?"
for p in split mixed f16; do
  timeout 900 bash tools/probe/profile_split.sh $p gpurun_out/r04/prof_$p > gpurun_out/r04/prof_$p.log 2>&1; echo "prof $p rc=$?"
  python3 tools/pmc_summary.py gpurun_out/r04/prof_$p profiles/r04_pmc_$p.txt $p > /dev/null
  cp gpurun_out/r04/prof_$p/kt/*kernel_stats.csv profiles/r04_kernel_stats_$p.csv
  cp profiles/r04_pmc_$p.txt profiles/r04_kernel_stats_$p.csv profiles/pmc_traffic.json $P/
  rm -rf gpurun_out/r04/prof_$p
done
timeout 900 python bench.py > profiles/r04_bench.json 2> gpurun_out/r04/bench.err; echo "bench rc=$?"; cp profiles/r04_bench.json $P/
(python tests/tools/model_bench.py f16 split f32 2>&1 | grep model; python tests/tools/model_bench.py --hetero f16 split 2>&1 | grep model;
 python tests/tools/encoder_bench.py 2>&1 | grep PointPillar;
 python tests/tools/camera_bench.py f16 split f32 2>&1 | grep Cvt) > $P/r04_model.txt; echo "model rc=$?"
(echo "# PointPillar encoder, per-launch kernel durations (tools/probe/r04_conv_layers.sh)"; bash tools/probe/r04_conv_layers.sh 2>&1 | grep -v "^\[") > $P/r04_conv_layers.txt
(echo "# CVT camera encoder (split), kernels per forward (tools/probe/r04_cvt_layers.sh)"; bash tools/probe/r04_cvt_layers.sh split 2>&1) > $P/r04_cvt_layers.txt
cut -c1-400 $P/r04_bench.json; cat $P/r04_model.txt
