"""Which weights, scaled by 100, make the training forward / backward of the fusion non-finite? (training-range item)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import hmvit_oracle as O
C, w, L, H, W = (256, 8, 3, 16, 24) if "256" in sys.argv else (64, 4, 3, 16, 24)
cfg = O.make_config(C, w, L, voxel=0.4, downsample=4)
scene = O.synthetic_scene(L, C, H, W, [0, 1, 0], n_valid=3, seed=6, tx_step=3.0, ty_step=-2.0)
gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(7)).cuda()
groups = {"q_linears": ["q_linears"], "k_linears": ["k_linears"], "v_linears": ["v_linears"], "a_linears": ["a_linears"],
          "ffn": [".fn.net."], "mlp_head": ["mlp_head"], "all": ["linears", ".fn.net.", "mlp_head"]}
for scale in (30.0, 100.0):
    for gname, pats in groups.items():
        sd = O.random_state_dict(cfg, seed=5)
        for k in sd:
            if any(p in k for p in pats) and sd[k].is_floating_point():
                sd[k] = sd[k] * scale
        net = hmvit_amd.HeteroFusion(cfg, precision="f32")
        net.load_state_dict(sd, strict=True)
        net = net.cuda().eval()
        x = scene[0].cuda().requires_grad_(True)
        y = net(x, *[t.cuda() for t in scene[1:]])
        (y * gy).sum().backward()
        bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print(f"x{scale:g} {gname}: forward finite {bool(torch.isfinite(y).all())} max|y| {float(y.abs().max()):.2e}; d/dx finite "
              f"{bool(torch.isfinite(x.grad).all())}; non-finite parameter gradients: {len(bad)} {bad[:3]}")
