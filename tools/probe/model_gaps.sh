# kernel-trace of the whole LiDAR model at the shipped size: launches, busy time and gaps of the last forward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
PREC=${1:-split}
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r02/kt_model -o kt -- python3 tests/tools/model_bench.py $PREC > gpurun_out/r02/kt_model.log 2>&1
grep model gpurun_out/r02/kt_model.log
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/r02/kt_model/*kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_pfn_scatter" in r["Kernel_Name"]]
a=idx[-1]; seg=rows[a:]
t0=int(seg[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in seg)
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)
print("launches",len(seg),"span ms",(t1-t0)/1e6,"busy ms",busy/1e6)
gaps=[(int(seg[i+1]["Start_Timestamp"])-int(seg[i]["End_Timestamp"]))/1e3 for i in range(len(seg)-1)]
big=sorted([(round(g,1),seg[i]["Kernel_Name"][:44],seg[i+1]["Kernel_Name"][:44]) for i,g in enumerate(gaps)],reverse=True)[:8]
for b in big: print(b)
PY
