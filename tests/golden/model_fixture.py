"""Seeded config / weights / batch of the end-to-end model golden (g9).  Shared by the golden
generator (build container, imports the reference) and the tests (GPU box, no reference): only numpy
legacy-stream generators from oracle/ are used, so both sides rebuild identical tensors."""
import numpy as np
import torch

from oracle import decoder_oracle as DO
from oracle import hmvit_oracle as O
from oracle import pointpillar_oracle as PO


def model_config():
    """Small LiDAR-only HM-ViT config: 64x48 pillar canvas -> 12x16 BEV, window 4, L = 3."""
    largs = PO.make_args(64, 48)
    fus = O.make_config(256, 4, 3, voxel=0.4, downsample=4)
    st = {"downsample_rate": 4, "voxel_size": [0.4, 0.4, 4], "use_roi_mask": True}
    return {"max_cav": 3, "anchor_number": 2, "compression": 0, "spatial_transform": st, "camera": {},
            "lidar": largs, "hetero_fusion": fus, "hetero_decoder": DO.make_params()}


def model_state_dict(cfg, seed):
    sd = {}
    sd.update({f"lidar_encoder.{k}": v for k, v in PO.random_state_dict(cfg["lidar"], seed).items()})
    sd.update({f"fusion_net.{k}": v for k, v in O.random_state_dict(cfg["hetero_fusion"], seed + 1).items()})
    sd.update(DO.random_state_dict(cfg["hetero_decoder"], seed + 2, prefix="decoder"))
    rs = np.random.RandomState(seed + 3)
    for name, co in (("cls_head", 2), ("reg_head", 14)):
        sd[f"{name}.weight"] = torch.from_numpy(rs.uniform(-0.06, 0.06, (co, 256, 1, 1)).astype(np.float32))
        sd[f"{name}.bias"] = torch.from_numpy(rs.uniform(-0.06, 0.06, co).astype(np.float32))
    return sd


def model_batch(cfg, seed):
    """B = 2, record_len [3, 2], all LiDAR; 5 agents x 300 pillars."""
    vf, vc, vn = PO.synthetic_pillars(5, 300, 64, 48, cfg["lidar"], seed=seed)
    _, pw, _, _, _ = O.synthetic_scene(3, 1, 1, 1, [1, 1, 1], seed=0, B=2, tx_step=3.0, ty_step=-2.0)
    eye = torch.eye(4)
    pw[1, 2, :] = eye
    pw[1, :, 2] = eye
    mode = torch.tensor([[1.0, 1.0, 1.0], [1.0, 1.0, 0.0]], dtype=torch.float64)   # zero-padded floats as collate_batch emits
    return {"mode": mode, "record_len": torch.tensor([3, 2]), "pairwise_t_matrix": pw,
            "processed_lidar": {"voxel_features": vf, "voxel_coords": vc, "voxel_num_points": vn}}


