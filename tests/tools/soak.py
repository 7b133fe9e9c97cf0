"""Soak run: the cfg2 forward repeated N times on the same inputs; every output must be bit-identical to the first one
(a race in the LDS-DMA ring, the producer / consumer barriers or the staged stores would show up as a rare mismatch)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmvit_amd import synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for name, (L, H, W, modes) in {"cfg2": (5, 200, 704, [1] * 5), "mixed 3 agents": (3, 96, 160, [1, 0, 1])}.items():
    cfg = S.make_config(256, 8, L, voxel=0.4, downsample=1)
    net = S.seeded_fusion(cfg, "f16").cuda().eval()
    scene = [t.cuda() for t in S.synthetic_scene(L, 256, H, W, modes, seed=1)]
    ref = net(*scene).clone()
    bad, t0 = 0, time.time()
    for i in range(n):
        y = net(*scene)
        if i % 10 == 9 and not torch.equal(y, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"{name}: {n} forwards, {n // 10} compared, {bad} mismatches, {1e3 * (time.time() - t0) / n:.2f} ms per forward, finite {bool(torch.isfinite(ref).all())}")

# the whole LiDAR-only model (pillars -> heads): convolution kernels included
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden"))
import hmvit_amd
from hmvit_amd import replay as R
cfg = R.lidar_model_config(512, 512, max_cav=5)
torch.manual_seed(0)
model = hmvit_amd.BevformerPointPillarHetero(cfg, precision="f16").cuda().eval()
pre = hmvit_amd.SpVoxelPreprocessor(R.preprocess_params(cfg), train=False)
frame = R.SyntheticReplayDataset(cfg, 1, seed=3)[0]
lidar = pre.collate_batch([pre.preprocess(c) for c in frame["clouds"]])
batch = {"mode": frame["mode"], "record_len": frame["record_len"], "pairwise_t_matrix": frame["pairwise_t_matrix"].cuda(), "processed_lidar": lidar}
ref = model(batch)
ref = (ref["psm"].clone(), ref["rm"].clone())
m = max(50, n // 5)
bad = 0
for i in range(m):
    out = model(batch)
    if i % 5 == 4 and not (torch.equal(out["psm"], ref[0]) and torch.equal(out["rm"], ref[1])):
        bad += 1
torch.cuda.synchronize()
print(f"LiDAR-only model, shipped size: {m} forwards, {m // 5} compared, {bad} mismatches, finite {bool(torch.isfinite(ref[0]).all() and torch.isfinite(ref[1]).all())}")
