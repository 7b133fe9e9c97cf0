"""Where the f16 PointPillar encoder's error comes from: the shipped-size encoder (5 agents x 20 k pillars,
512 x 512 grid, layer_nums [3, 5, 8]) in f16 mode against the same kernels in exact-f32 mode, stage by stage
(PointPillar.trace), rel-max = max|a - b| / max|b| per stage.  Two seeds: default-initialised weights (random_state_dict)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import pointpillar_oracle as PO

n_agents, nx, ny = 5, 512, 512
args = PO.make_args(nx, ny, small=False)
for seed in (1, 3):
    sd = PO.random_state_dict(args, seed=seed)
    vf, vc, vn = PO.synthetic_pillars(n_agents, 20000, nx, ny, args, seed=seed + 1)
    batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()},
             "record_len": torch.tensor([n_agents])}
    traces = {}
    for prec in ("f32", "f16"):
        net = hmvit_amd.PointPillar(args, precision=prec)
        net.load_state_dict(sd, strict=False)
        net = net.cuda().eval()
        net.set_return_features()
        net.trace = []
        net(batch)
        torch.cuda.synchronize()
        traces[prec] = net.trace
    print(f"seed {seed}:")
    for (name, a), (_, b) in zip(traces["f16"], traces["f32"]):
        err = float((a - b).abs().max() / b.abs().max())
        rms = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
        print(f"  {name:20s} rel-max {err:.2e}   rel-rms {rms:.2e}   max|ref| {float(b.abs().max()):.3g}")
