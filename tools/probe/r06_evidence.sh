# Round-6 evidence on the final sources: the whole GPU suite with durations, smoke(), kernel-trace + PMC profiles of the three bench
# modes (separate passes), the default bench line (roofline.traffic from the profiles of the same sources), per-stage attention figures,
# encoder traffic, training step + its kernel stats, model / encoder side figures, the pcs2 and x16 cycle traces (probe library).
# Every artefact is copied into gpurun_out/r06/profiles/ as soon as it exists (a call that is cut off keeps what it has).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r06/profiles; mkdir -p $P
timeout 1500 python -m pytest tests -m gpu -q -s --durations=60 -p no:cacheprovider > gpurun_out/r06/gputest_final.log 2>&1; echo "rc=$?" >> gpurun_out/r06/gputest_final.log
grep -E "FAILED|ERROR|passed|failed|rc=" gpurun_out/r06/gputest_final.log | head -20
grep -E "^range\[|^train range|^64x176|^   d/dx|backward run-to-run|float64 oracle, CPU vs GPU" gpurun_out/r06/gputest_final.log > $P/r06_range.txt
(echo "# pytest -m gpu --durations=60 on the final sources of round 6 (one MI355X box, 256-thread EPYC host)"; sed -n '/slowest/,$p' gpurun_out/r06/gputest_final.log | head -75) > $P/r06_gputest_durations.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r06/smoke.log
for p in split mixed f16; do
  timeout 900 bash tools/probe/profile_split.sh $p gpurun_out/r06/prof_$p > gpurun_out/r06/prof_$p.log 2>&1; echo "prof $p rc=$?"
  python3 tools/pmc_summary.py gpurun_out/r06/prof_$p profiles/r06_pmc_$p.txt $p > /dev/null
  cp gpurun_out/r06/prof_$p/kt/*kernel_stats.csv profiles/r06_kernel_stats_$p.csv
  cp profiles/r06_pmc_$p.txt profiles/r06_kernel_stats_$p.csv profiles/pmc_traffic.json $P/
  rm -rf gpurun_out/r06/prof_$p
done
timeout 600 bash tools/probe/r06_encoder_traffic.sh > gpurun_out/r06/enc_traffic.log 2>&1; echo "encoder traffic rc=$?"; cp profiles/pmc_traffic.json $P/
timeout 900 python bench.py > profiles/r06_bench.json 2> gpurun_out/r06/bench.err; echo "bench rc=$?"; cp profiles/r06_bench.json $P/
timeout 600 bash tools/probe/r06_attn_prof.sh final > $P/r06_attention_per_stage.txt 2>&1; echo "attn per stage rc=$?"
(python tests/tools/train_bench.py native 3 2>&1 | tail -1; python tests/tools/train_bench.py cfg2 3 2>&1 | tail -1; python bench.py --train --steps 3 --warmup 1 2>/dev/null | tail -1) > $P/r06_train.txt; echo "train rc=$?"
timeout 900 bash tools/probe/r06_train_pmc.sh > gpurun_out/r06/train_pmc.log 2>&1; cp gpurun_out/r06/train_kernel_stats.csv $P/r06_train_kernel_stats.csv; cp gpurun_out/r06/train_pmc.txt $P/r06_train_pmc.txt; echo "train pmc rc=$?"
(python tests/tools/model_bench.py f16 split f32 2>&1 | grep model; python tests/tools/model_bench.py --hetero f16 split 2>&1 | grep model;
 python tests/tools/encoder_bench.py 2>&1 | grep PointPillar;
 python tests/tools/camera_bench.py f16 split f32 2>&1 | grep Cvt) > $P/r06_model.txt; echo "model rc=$?"
python tests/tools/ap_replay.py --scenes 24 --precision split > $P/r06_ap_replay.json 2>/dev/null; echo "ap rc=$?"
HMVIT_LIB=tools/probe/lib_probe.so python tests/tools/pcs2_trace.py > $P/r06_pcs2_trace.txt 2>&1
HMVIT_PATCH_ATTENTION=1 HMVIT_LIB=tools/probe/lib_probe.so python tests/tools/patch_trace.py > $P/r06_patch_trace.txt 2>&1
timeout 600 bash tools/probe/r06_ta_pmc.sh > $P/r06_ta_pmc.txt 2>&1; echo "ta pmc rc=$?"
HMVIT_PATCH_ATTENTION=1 timeout 600 bash tools/probe/r06_ta_pmc.sh > $P/r06_ta_pmc_patch.txt 2>&1; echo "ta pmc (patch) rc=$?"
HMVIT_PATCH_ATTENTION=1 timeout 600 bash tools/probe/r06_attn_prof.sh patch > $P/r06_attention_per_stage_patch.txt 2>&1; echo "attn per stage (patch) rc=$?"
HMVIT_LIB=tools/probe/lib_probe.so python tools/probe/x16_trace.py split > $P/r06_x16_trace.txt 2>&1
HMVIT_PATCH_ATTENTION=2 HMVIT_LIB=tools/probe/lib_probe.so python tests/tools/patch16_trace.py > $P/r06_patch16_trace.txt 2>&1
(echo "# tools/probe/r06_tailpull.sh: x16 tails pulled (static=0) against one workgroup per tile (static=1), probe library, interleaved on one box"; bash tools/probe/r06_tailpull.sh "pulled_tail or (full_size and split)" 2>&1 | grep -E "passed|failed|static=") > $P/r06_tailpull_ab.txt
(echo "# tools/probe/r06_soak_pull.py 300: pulled tiles, every forward bit-compared with the first and with one workgroup per tile"; python tools/probe/r06_soak_pull.py 300 2>&1 | grep -E "forwards") > $P/r06_soak_pull.txt
(echo "# tools/probe/r06_clocks.sh split: rocm-smi once a second while bench.py --precision split --steps 5000 loops"; bash tools/probe/r06_clocks.sh split 2>&1 | sed 's/=\{5,\}//g; s/GPU\[0\]\t*: //g; s/Power Consumption//' | grep -E "sclk|scenes") > $P/r06_clocks.txt
cut -c1-600 $P/r06_bench.json; cat $P/r06_model.txt $P/r06_train.txt $P/r06_attention_per_stage.txt
