cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict"
rm -rf gpurun_out/prof2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/prof2/fetch -o p -- $B > gpurun_out/prof2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -f csv -d gpurun_out/prof2/tc -o p -- $B > gpurun_out/prof2_tc.log 2>&1
python3 - <<PY
import csv, collections
for f in ("gpurun_out/prof2/fetch/p_counter_collection.csv", "gpurun_out/prof2/tc/p_counter_collection.csv"):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_attention_pc" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(k, ["%.3e"%x for x in v[:8]])
PY
