"""Times the PointPillar BEV encoder (rows a14-a16) at the shipped size: 5 agents, 20 k pillars each on a
512 x 512 grid, layer_nums [3, 5, 8] -> (5, 256, 128, 128).  Not the headline bench; a tuning aid."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import pointpillar_oracle as PO

n_agents, nx, ny = 5, 512, 512
args = PO.make_args(nx, ny, small=False)
sd = PO.random_state_dict(args, seed=1)
vf, vc, vn = PO.synthetic_pillars(n_agents, 20000, nx, ny, args, seed=2)
batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()},
         "record_len": torch.tensor([n_agents])}
for prec in ("f16", "split", "f32"):
    net = hmvit_amd.PointPillar(args, precision=prec)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    net.set_return_features()
    torch.cuda.empty_cache()
    for _ in range(4):                                            # warm-up: weight preparation, allocator growth
        y = net(batch)
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = net(batch)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"PointPillar {prec}: {ms:.2f} ms per 5-agent call, out {tuple(y.shape)}, {0.1436 * n_agents / (ms / 1e3):.1f} TFLOP/s on the convs (143.6 GF/agent)")
