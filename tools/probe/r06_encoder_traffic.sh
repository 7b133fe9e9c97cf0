# HBM bytes per forward of the two encoders in split mode (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes, summed over
# every kernel of a forward; FETCH doubled per the gfx950 wide-read note) -> profiles/pmc_traffic.json["encoders_split"], stamped with
# the kernel-source hash like the fusion figures: bench.py attaches them to encoders.*.traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06/enc_traffic; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/pp.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import hmvit_amd
from hmvit_amd import synthetic as S, replay as R
which = sys.argv[1]
L = 5
if which == "pointpillar":
    mcfg = R.lidar_model_config(512, 512, max_cav=L)
    vf, vc, vn = S.synthetic_pillars(L, 20000, 512, 512, mcfg["lidar"], seed=2)
    batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}, "record_len": torch.tensor([L]), "n_agents": L}
    net = hmvit_amd.PointPillar(mcfg["lidar"], precision="split").cuda().eval(); net.set_return_features()
else:
    from hmvit_amd.camera import CvtCameraEncoder
    ccfg = S.camera_config(image=512, num_layers=34, bev_h=256, bev_w=256)
    batch = {k: v.cuda() for k, v in S.synthetic_cameras(L, 512, seed=8).items()}
    net = CvtCameraEncoder(ccfg, precision="split").cuda().eval()
with torch.no_grad():
    for _ in range(3): net(batch)          # weight preparation, allocator
    torch.cuda.synchronize()
    print("MARK", flush=True)
    for _ in range(4): net(batch)
    torch.cuda.synchronize()
PY
for w in pointpillar cvt; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/${w}_$c -o p -- python3 $OUT/pp.py $w > $OUT/${w}_$c.log 2>&1
  done
done
python3 - <<PY
import csv, glob, json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
res = {}
for w in ("pointpillar", "cvt"):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob("$OUT/%s_%s/*counter_collection.csv" % (w, c)): rows += list(csv.DictReader(open(f)))
        rows = [r for r in rows if r["Counter_Name"] == c]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        # 7 forwards; a kernel that runs once per forward marks the cycle (k_pfn_scatter / k_maxpool): from its 4th occurrence to
        # the end are exactly four cycles of the forward's dispatch sequence (weight preparation sits in the first cycle)
        marker = "k_pfn_scatter" if w == "pointpillar" else "k_maxpool"
        marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
        assert len(marks) == 7, (w, c, len(marks))
        per = [sum(float(r["Counter_Value"]) for r in rows[marks[j]:(marks[j + 1] if j + 1 < 7 else len(rows))]) for j in range(3, 7)]
        # the open-ended last cycle lacks the dispatches that precede its marker in program order, the cycles 3..5 are complete
        tot[c] = sorted(per[:3])[1] * 1024
    res[w] = int(2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"])
    print(w, "FETCH x2 %.3f GB, WRITE %.3f GB per forward" % (2 * tot["FETCH_SIZE"] / 1e9, tot["WRITE_SIZE"] / 1e9))
path = os.path.join(os.environ["GRAFT_REPO_ROOT"], "profiles", "pmc_traffic.json")
allt = json.load(open(path)) if os.path.exists(path) else {}
h = bench.kernel_source_hash()
if allt.get("kernel_source_hash") != h: allt = {"kernel_source_hash": h}
allt["encoders_split"] = res
json.dump(allt, open(path, "w"), indent=1)
PY
mkdir -p gpurun_out/r06/profiles; cp profiles/pmc_traffic.json gpurun_out/r06/profiles/
rm -rf $OUT/*_SIZE
