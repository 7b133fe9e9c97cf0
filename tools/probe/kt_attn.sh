cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt2
rocprofv3 --kernel-trace -f csv -d gpurun_out/kt2 -o kt -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict > gpurun_out/kt2.log 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/kt2/kt_kernel_trace.csv"))]
seq=[(r["Kernel_Name"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if "hmvit" in r["Kernel_Name"]]
# last forward: 
names=[n.split("(")[0][-40:] for n,_ in seq]
last=seq[-14:]
for n,t in last: print(f"{n.split('(')[0][-48:]:50s} {t:8.1f} us")
PY
