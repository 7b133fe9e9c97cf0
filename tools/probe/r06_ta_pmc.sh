# vector-memory path counters of the split attention launches (separate PMC passes; kernel-trace only beside them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06/ta_pmc; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --precision split --steps 4 --warmup 1 --no-cpu-baseline --no-strict"
i=0
# (a pass with TA_TA_BUSY_sum / TA_BUFFER_TOTAL_CYCLES_sum / TA_*_STALLED_BY_TC_CYCLES_sum aborted inside rocprofv3 and hung in its
# finaliser for 20 minutes on this pool: not collected; every pass runs under its own timeout)
for C in "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/p$i -o p -- $B > $OUT/p$i.log 2>&1; echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    rs = [r for r in csv.DictReader(open(f)) if "k_attention_p" in r["Kernel_Name"] or "k_out_ffn_qkv16" in r["Kernel_Name"]]
    for kn in ("k_attention_p", "k_out_ffn_qkv16"):
        ids = sorted({int(r["Dispatch_Id"]) for r in rs if kn in r["Kernel_Name"]})
        n = 4 if kn == "k_attention_p" else 3
        pos = {d: i % n for i, d in enumerate(ids)}
        for r in rs:
            if kn in r["Kernel_Name"]:
                acc[(kn, r["Counter_Name"])][pos[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
for (kn, cn), st in sorted(acc.items()):
    print(f"{kn:18s} {cn:40s}", {k: f"{sorted(v)[len(v) // 2]:.3e}" for k, v in sorted(st.items())})
PY
