"""Split-mode soak: the cfg2 forward (pulled item assignment in the dilated-grid launches) and a mixed-type scene repeated on the same
inputs, every output compared bit for bit with the first; then the LiDAR-only model in split (ring convolutions, range slots)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import synthetic as S, replay as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for name, (L, H, W, modes) in {"cfg2": (5, 200, 704, [1] * 5), "cfg3 types 10110": (5, 200, 704, [1, 0, 1, 1, 0]), "3 agents 96x160": (3, 96, 160, [1, 0, 1])}.items():
    cfg = S.make_config(256, 8, L, voxel=0.4, downsample=1)
    net = S.seeded_fusion(cfg, "split").cuda().eval()
    scene = [t.cuda() for t in S.synthetic_scene(L, 256, H, W, modes, seed=1)]
    with torch.no_grad():
        ref = net(*scene).clone()
        bad, t0 = 0, time.time()
        for i in range(n):
            y = net(*scene)
            if not torch.equal(y, ref):
                bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {n} forwards, all compared, {bad} mismatches, {1e3 * (time.time() - t0) / n:.2f} ms per forward, finite {bool(torch.isfinite(ref).all())}")
cfg = R.lidar_model_config(512, 512, max_cav=5)
torch.manual_seed(0)
model = hmvit_amd.BevformerPointPillarHetero(cfg, precision="split").cuda().eval()
pre = hmvit_amd.SpVoxelPreprocessor(R.preprocess_params(cfg), train=False)
frame = R.SyntheticReplayDataset(cfg, 1, seed=3)[0]
lidar = pre.collate_batch([pre.preprocess(c) for c in frame["clouds"]])
batch = {"mode": frame["mode"], "record_len": frame["record_len"], "pairwise_t_matrix": frame["pairwise_t_matrix"].cuda(), "processed_lidar": lidar}
with torch.no_grad():
    ref = model(batch)
    ref = (ref["psm"].clone(), ref["rm"].clone())
    m, bad = max(50, n // 4), 0
    for i in range(m):
        out = model(batch)
        if not (torch.equal(out["psm"], ref[0]) and torch.equal(out["rm"], ref[1])):
            bad += 1
print(f"LiDAR model split: {m} forwards, all compared, {bad} mismatches")
