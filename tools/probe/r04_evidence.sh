# Round-4 evidence run: the GPU suite, smoke(), kernel-trace + PMC profiles of the three bench modes (summaries into profiles/ so
# that the bench line that follows carries roofline.traffic of the very sources it runs), the default bench line, AP replay, trainer.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/profiles
timeout 3000 python -m pytest tests -m gpu -q -s > gpurun_out/r04/gputest.log 2>&1; echo "rc=$?" >> gpurun_out/r04/gputest.log
grep -E "FAILED|ERROR|passed|failed|rc=" gpurun_out/r04/gputest.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.log 2>&1; echo "smoke rc=$?"
for p in split mixed f16; do
  timeout 900 bash tools/probe/profile_split.sh $p gpurun_out/r04/prof_$p > gpurun_out/r04/prof_$p.log 2>&1; echo "prof $p rc=$?"
  python3 tools/pmc_summary.py gpurun_out/r04/prof_$p profiles/r04_pmc_$p.txt $p > /dev/null
  cp gpurun_out/r04/prof_$p/kt/*kernel_stats.csv profiles/r04_kernel_stats_$p.csv
done
timeout 900 python bench.py > profiles/r04_bench.json 2> gpurun_out/r04/bench.err; echo "bench rc=$?"
for p in f32 split f16; do timeout 600 python tests/tools/ap_replay.py --precision $p 2>> gpurun_out/r04/ap_replay.err; done > profiles/r04_ap_replay.json; echo "ap rc=$?"
(python tests/tools/train_bench.py native 5 2>&1 | tail -1; python tests/tools/train_bench.py cfg2 3 2>&1 | tail -1;
 python -m hmvit_amd.trainer --epochs 2 --frames 6 --agents 5 --grid 512 192 2>/dev/null | tail -1;
 python -m hmvit_amd.trainer --epochs 2 --frames 6 --agents 5 --grid 512 192 --train_lidar_backbone 2>/dev/null | tail -1;
 python -m hmvit_amd.trainer --epochs 2 --frames 6 --agents 5 --grid 256 128 --camera_ratio 0.5 --val_frames 2 2>/dev/null | tail -1;
 python bench.py --train --steps 3 --warmup 1 2>/dev/null | tail -1) > profiles/r04_train.txt; echo "train rc=$?"
(python tests/tools/model_bench.py f16 split f32 2>&1 | grep model; python tests/tools/model_bench.py --hetero f16 split 2>&1 | grep model;
 python tests/tools/encoder_bench.py 2>&1 | grep PointPillar;
 python tests/tools/camera_bench.py f16 split f32 2>&1 | grep Cvt) > profiles/r04_model.txt; echo "model rc=$?"
bash tools/probe/r04_attn_prof.sh final > profiles/r04_attention_per_stage.txt 2>&1
(echo "# PointPillar encoder, per-launch kernel durations (tools/probe/r04_conv_layers.sh)"; bash tools/probe/r04_conv_layers.sh 2>&1 | grep -v "^\[") > profiles/r04_conv_layers.txt
(echo "# CVT camera encoder (split), kernels per forward (tools/probe/r04_cvt_layers.sh)"; bash tools/probe/r04_cvt_layers.sh split 2>&1) > profiles/r04_cvt_layers.txt
(echo "# finish times of the 256 persistent attention workgroups, dilated-grid launch: pulled (shipped) vs static item assignment (probe build, tools/probe/r04_attn_balance.py)";
 HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so python tools/probe/r04_attn_balance.py 2>&1 | tail -2;
 HMVIT_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_probe.so HMVIT_PCS_STATIC=1 python tools/probe/r04_attn_balance.py 2>&1 | tail -2) > profiles/r04_attn_balance.txt
bash tools/probe/r04_train_pmc.sh > gpurun_out/r04/train_pmc.log 2>&1; cp gpurun_out/r04/train_pmc.txt profiles/r04_train_pmc.txt; cp gpurun_out/r04/train_kernel_stats.csv profiles/r04_train_kernel_stats.csv
grep -E "^range\[" gpurun_out/r04/gputest.log > profiles/r04_range.txt
cp profiles/r04_* profiles/pmc_traffic.json gpurun_out/r04/profiles/
cut -c1-400 profiles/r04_bench.json; cat profiles/r04_train.txt profiles/r04_model.txt
