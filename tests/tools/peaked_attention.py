"""How the precision modes behave when the attention sharpens: the query projection of every stage is scaled up (logits grow
proportionally, the softmax approaches a one-hot selection of a key), rel-max error of each mode against the oracle."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import hmvit_oracle as O
cfg = O.make_config(256, 8, 3, voxel=0.4, downsample=4)
for scale in (1.0, 16.0, 64.0, 256.0, 1024.0):
    sd = O.random_state_dict(cfg, seed=7)
    for k in list(sd):
        if "q_linears" in k:
            sd[k] = sd[k] * scale
    scene = O.synthetic_scene(3, 256, 32, 48, [1, 0, 1], n_valid=3, seed=3, tx_step=6.0, ty_step=-4.0)
    ref = O.hetero_fusion(*scene, sd, cfg)
    out = {}
    for prec in ("f32", "split", "mixed", "f16"):
        net = hmvit_amd.HeteroFusion(cfg, precision=prec); net.load_state_dict(sd, strict=True); net = net.cuda().eval()
        y = net(*[t.cuda() for t in scene]).cpu()
        out[prec] = float((y - ref).abs().max() / ref.abs().max())
    print(scale, {k: f"{v:.2e}" for k, v in out.items()})
