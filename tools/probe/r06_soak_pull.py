"""Round-6 soak of the pulled tiles (x16 stage tails and the first stage's k_ln_qkv16: tail16_pull): split and mixed modes, cfg2, a mixed-type
scene (job classes: segments), a two-sample ragged scene with strongly shifted poses (dead tiles) - every forward compared bit for bit with the
first one AND with the one-workgroup-per-tile launch form (skip_masked = 0)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmvit_amd import synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cases = {"cfg2": dict(L=5, H=200, W=704, modes=[1] * 5), "cfg3 types 10110": dict(L=5, H=200, W=704, modes=[1, 0, 1, 1, 0]),
         "2 samples 40x56 shifted": dict(L=4, H=40, W=56, modes=[1, 0, 0, 1], B=2, yaw_step=0.45, tx_step=30.0, ty_step=-20.0),
         "3 agents 96x160": dict(L=3, H=96, W=160, modes=[1, 0, 1])}
for prec in ("split", "mixed"):
    for name, c in cases.items():
        c = dict(c)
        L, H, W, modes = c.pop("L"), c.pop("H"), c.pop("W"), c.pop("modes")
        cfg = S.make_config(256, 8, L, voxel=0.4, downsample=1)
        net = S.seeded_fusion(cfg, prec).cuda().eval()
        scene = [t.cuda() for t in S.synthetic_scene(L, 256, H, W, modes, seed=1, **c)]
        with torch.no_grad():
            net.skip_masked = 0
            static = net(*scene).clone()
            net.skip_masked = 1
            ref = net(*scene).clone()
            bad, t0 = 0, time.time()
            for i in range(n if H * W > 20000 else 3 * n):
                if not torch.equal(net(*scene), ref):
                    bad += 1
        torch.cuda.synchronize()
        print(f"{prec} {name}: {n if H * W > 20000 else 3 * n} forwards, {bad} mismatches, equal to one workgroup per tile: {bool(torch.equal(ref, static))}, "
              f"finite {bool(torch.isfinite(ref).all())}, {1e3 * (time.time() - t0) / (n if H * W > 20000 else 3 * n):.2f} ms per forward")
