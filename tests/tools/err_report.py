"""Prints rel-max errors of the f16 path against the oracle / goldens for a few scenes (tuning aid, not a test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import hmvit_amd
from oracle import hmvit_oracle as O

def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())

def fusion(cfg, sd, precision):
    net = hmvit_amd.HeteroFusion(cfg, precision=precision).cuda()
    net.load_state_dict(sd)
    return net

cases = []
for modes, nv in [([1, 1, 1, 1, 1], 5), ([0, 1, 1, 0, 1], 4), ([0, 0, 0, 0, 0], 5)]:
    cfg = O.make_config(256, 8, 5, voxel=0.4, downsample=4)
    sd = O.random_state_dict(cfg, seed=7)
    scene = O.synthetic_scene(5, 256, 32, 48, modes, n_valid=nv, seed=3, tx_step=6.0, ty_step=-4.0)
    cases.append((f"win8 {modes}", cfg, sd, scene))
cfg = O.make_config(256, 8, 4, voxel=0.4, downsample=4, arch="parallel")
cases.append(("parallel", cfg, O.random_state_dict(cfg, seed=13), O.synthetic_scene(4, 256, 32, 32, [1, 0, 0, 1], n_valid=3, seed=6, tx_step=5.0, ty_step=-3.0)))
cfg = O.make_config(256, 8, 5, voxel=0.4, downsample=4)
cases.append(("win8 64x96 yaw", cfg, O.random_state_dict(cfg, seed=23), O.synthetic_scene(5, 256, 64, 96, [1, 0, 1, 1, 0], seed=11, yaw_step=0.35, tx_step=4.0, ty_step=3.0)))
for name, cfg, sd, scene in cases:
    ref = O.hetero_fusion(*scene, sd, cfg)
    dev = [t.cuda() for t in scene]
    out = {}
    for prec in ("f32", "f16"):
        out[prec] = fusion(cfg, sd, prec)(*dev).cpu()
    print(f"{name:28s} f32 {rel(out['f32'], ref):.2e}  f16 {rel(out['f16'], ref):.2e}", flush=True)
