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
