"""Times the whole LiDAR-only HM-ViT model at the shipped size: 5 agents x 20 k pillars on a 512 x 512 grid -> PointPillar ->
(5, 256, 128, 128) BEV maps -> HeteroFusion (window 8, 2 iterations) -> HeteroDecoder -> cls / reg heads, then the
post-processor (decode + rotated NMS).  Random weights, synthetic pillars; a tuning aid, not the headline bench."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import hmvit_amd
from hmvit_amd import synthetic as S
from model_fixture import model_state_dict
from oracle import decoder_oracle as DO
from oracle import pointpillar_oracle as PO

L, nx, ny = 5, 512, 512
largs = PO.make_args(nx, ny, small=False)
st = {"downsample_rate": 4, "voxel_size": [0.4, 0.4, 4], "use_roi_mask": True}
cfg = {"max_cav": L, "anchor_number": 2, "compression": 0, "spatial_transform": st, "camera": {}, "lidar": largs,
       "hetero_fusion": S.make_config(256, 8, L, voxel=0.4, downsample=4), "hetero_decoder": DO.make_params()}
sd = model_state_dict(cfg, 5)
vf, vc, vn = PO.synthetic_pillars(L, 20000, nx, ny, largs, seed=2)
_, pw, _, _, _ = S.synthetic_scene(L, 1, 1, 1, [1] * L, seed=0)
batch = {"mode": torch.ones(1, L, dtype=torch.float64), "record_len": torch.tensor([L]), "pairwise_t_matrix": pw.cuda(),
         "processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}}
from hmvit_amd.camera import CvtCameraEncoder
from oracle import camera_oracle as CAM
ccfg = CAM.make_config(image=512, num_layers=34, dim=128, bev=256)
csd = CAM.random_state_dict(ccfg, seed=7)
cams = {k: v.cuda() for k, v in CAM.synthetic_batch(L, ccfg, seed=8).items()}
hetero = "--hetero" in sys.argv
if hetero:   # BASELINE configs[2]: camera ego, mixed agent types; every agent carries both sensors' inputs, `mode` picks one
    batch["mode"] = torch.tensor([[0.0, 1.0, 0.0, 1.0, 1.0]], dtype=torch.float64)
    batch.update({"camera": cams["camera"], "intrinsic": cams["intrinsic"], "extrinsic": cams["extrinsic"],
                  "cav2cam_extrinsic": cams["extrinsic"].clone()})
for prec in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["f16"]):
    cam_enc = CvtCameraEncoder(ccfg, precision=prec) if hetero else None
    net = hmvit_amd.BevformerPointPillarHetero(cfg, camera_encoder=cam_enc, precision=prec)
    full = dict(sd)
    if hetero:
        full.update({f"camera_encoder.{k}": v for k, v in csd.items()})
    missing, unexpected = net.load_state_dict(full, strict=False)
    net = net.cuda().eval()
    out = net(batch); torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = net(batch)
    e1.record(); torch.cuda.synchronize()
    print(f"model {'hetero (2 camera + 3 LiDAR agents) ' if hetero else ''}{prec}: {e0.elapsed_time(e1) / n:.2f} ms per 5-agent scene (pillars -> psm {tuple(out['psm'].shape)}, rm {tuple(out['rm'].shape)})")
