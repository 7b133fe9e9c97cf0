# Round-2 evidence run: the GPU suite, smoke(), kernel-trace + PMC profiles of the three bench modes (summaries into profiles/ so
# that the bench line that follows carries roofline.traffic of the very sources it runs), the default bench line, AP replay, trainer.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/profiles
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r02/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r02/gputest.log
grep -E "FAILED|ERROR|passed|failed|rc=" gpurun_out/r02/gputest.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02/smoke.log 2>&1; echo "smoke rc=$?"
for p in split mixed f16; do
  timeout 900 bash tools/probe/profile_split.sh $p gpurun_out/r02/prof_$p > gpurun_out/r02/prof_$p.log 2>&1; echo "prof $p rc=$?"
  python3 tools/pmc_summary.py gpurun_out/r02/prof_$p profiles/r02_pmc_$p.txt $p > /dev/null
  cp gpurun_out/r02/prof_$p/kt/*kernel_stats.csv profiles/r02_kernel_stats_$p.csv
done
timeout 900 python bench.py > profiles/r02_bench.json 2> gpurun_out/r02/bench.err; echo "bench rc=$?"
for p in f32 split f16; do timeout 600 python tests/tools/ap_replay.py --precision $p 2>> gpurun_out/r02/ap_replay.err; done > profiles/r02_ap_replay.json; echo "ap rc=$?"
(python tests/tools/train_bench.py native 5 2>&1 | tail -1; python tests/tools/train_bench.py cfg2 3 2>&1 | tail -1;
 python -m hmvit_amd.trainer --epochs 2 --frames 6 --agents 5 --grid 512 192 2>/dev/null | tail -1;
 python -m hmvit_amd.trainer --epochs 2 --frames 6 --agents 5 --grid 512 192 --train_lidar_backbone 2>/dev/null | tail -1) > profiles/r02_train.txt; echo "train rc=$?"
(python tests/tools/model_bench.py f16 split f32 2>&1 | grep model; python tests/tools/model_bench.py --hetero f16 split 2>&1 | grep model;
 python tests/tools/encoder_bench.py 2>&1 | grep PointPillar;
 python tests/tools/camera_bench.py f16 split f32 2>&1 | grep Cvt) > profiles/r02_model.txt; echo "model rc=$?"
cp profiles/r02_* profiles/pmc_traffic.json gpurun_out/r02/profiles/
cut -c1-400 profiles/r02_bench.json; cat profiles/r02_train.txt profiles/r02_model.txt
