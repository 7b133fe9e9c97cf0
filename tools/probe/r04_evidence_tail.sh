# after a kernel change that leaves the fusion kernels alone: the affected tests, then the split profile + bench line again so that
# profiles/pmc_traffic.json carries the hash of the sources the bench line ran (bench.kernel_source_hash)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r04/profiles; mkdir -p $P
timeout 1500 python -m pytest tests/test_hip_range.py tests/test_hip_post.py tests/test_hip_ap.py tests/test_hip_trainer.py tests/test_hip_model.py tests/test_hip_encoder.py tests/test_hip_camera.py -m gpu -q 2>&1 | tail -3
for p in split; do
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
cut -c1-300 $P/r04_bench.json; cat $P/r04_model.txt
