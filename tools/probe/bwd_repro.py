"""Run-to-run reproducibility of the fusion's backward on the scene of test_backward_matches_oracle_autograd_five_agents_64x176
(or: L tx H W on the command line): the gradients of four passes, largest relative difference per tensor between passes.  The
shipped kernels differ only in the order of a few float atomics (1e-6); the round-3 experiments with the attention backward at two
workgroups per CU, and with a persistent producer / consumer form of it, differed by 1e-2 in one pass out of two."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import hmvit_oracle as O
Lc = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tx = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0
Hc, Wc = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (64, 176)
cfg = O.make_config(256, 8, Lc, voxel=0.4, downsample=4)
sd = O.random_state_dict(cfg, seed=7)
scene = O.synthetic_scene(Lc, 256, Hc, Wc, [1, 0, 1, 1, 0][:Lc], n_valid=Lc, seed=3, tx_step=tx, ty_step=-5.0 * tx / 9.0)
net = hmvit_amd.HeteroFusion(cfg, precision="f32")
net.load_state_dict(sd, strict=True)
net = net.cuda().eval()
net.force_autograd = True
gy = torch.randn(1, 256, Hc, Wc, generator=torch.Generator().manual_seed(5)).cuda()
def grads():
    net.zero_grad()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    out = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    out["x"] = x.grad.detach().clone()
    return out
runs = [grads() for _ in range(4)]
worst = {}
for r in runs[1:]:
    for k in r:
        worst[k] = max(worst.get(k, 0.0), float((r[k] - runs[0][k]).abs().max() / runs[0][k].abs().max().clamp_min(1e-30)))
top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
print("run to run, largest relative difference:", max(worst.values()), [(k.replace("hetero_fusion_block.", ""), f"{v:.1e}") for k, v in top])
