"""HeteroFusion.forward captured in a HIP graph through torch.cuda.graph (every launch of libhmvit goes to the current
stream, workspace and output come from torch's allocator): replay equals the eager result, and for small maps - where the
17 launches per scene, not the kernels, set the pace - it is faster."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hmvit_amd import synthetic as S

for name, (C, win, L, H, W, ds) in {"cfg1 (2 agents, 100x352, C=64, window 4)": (64, 4, 2, 100, 352, 2),
                                    "shipped 128x128 (5 agents, C=256)": (256, 8, 5, 128, 128, 4),
                                    "cfg2 (5 agents, 200x704, C=256)": (256, 8, 5, 200, 704, 1)}.items():
    cfg = S.make_config(C, win, L, voxel=0.4, downsample=ds)
    net = S.seeded_fusion(cfg, "f16").cuda().eval()
    scene = [t.cuda() for t in S.synthetic_scene(L, C, H, W, [1] * L, seed=1)]
    eager = net(*scene).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            net(*scene)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net(*scene)
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(out, eager)

    def timed(fn, n=200):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    te, tg = timed(lambda: net(*scene)), timed(g.replay)
    print(f"{name}: eager {te:.3f} ms, graph replay {tg:.3f} ms per scene, identical output: {same}")
