"""Procedurally generated "OPV2V synthetic replay" scenes shared by the checkpoint trainer (build container), the AP
replay tool and the GPU tests: vehicles of the anchor size in the ego frame, every agent (rigid pose T_i) sees points on the
vehicle surfaces in its own frame plus ground clutter; pairwise[i, j] = inv(T_j) T_i as the dataset builds it
(mixed/intermediate_fusion_dataset.py:163-202).  numpy legacy RandomState only, so every side regenerates identical scenes."""
import math

import numpy as np

from oracle import hmvit_oracle as O
from oracle import postprocess_oracle as PPO


def make_scene(rs, lidar_range, n_agents=3, n_obj=3, axis_aligned=True, pts_per_obj=220, n_ground=1200):
    """-> clouds [(n_i, 4) float32 per agent], pairwise (1, L, L, 4, 4), gt boxes (n_obj, 7) [x, y, z, h, w, l, yaw], gt corners."""
    rng = lidar_range
    poses = [O.rigid(0.15 * i, 2.5 * i, -1.5 * i) for i in range(n_agents)]          # agent i -> ego frame
    boxes = []
    tries = 0
    while len(boxes) < n_obj and tries < 1000:
        tries += 1
        yaw = (rs.choice([0.0, math.pi / 2]) + rs.normal(0, 0.04)) if axis_aligned else rs.uniform(-math.pi, math.pi)
        cand = [rs.uniform(rng[0] + 4, rng[3] - 4), rs.uniform(rng[1] + 4, rng[4] - 4), -1.0, 1.56, 1.6, 3.9, yaw]
        if all(math.hypot(cand[0] - b[0], cand[1] - b[1]) > 5.0 for b in boxes):
            boxes.append(cand)
    boxes = np.array(boxes, np.float32)
    gt = PPO.boxes_to_corners_3d(boxes, "hwl")
    clouds = []
    for i in range(n_agents):
        pts = []
        for b in boxes:
            n = pts_per_obj
            u = rs.uniform(-0.5, 0.5, (n, 3)) * np.array([b[5], b[4], b[3]])
            face = rs.randint(0, 3, n)
            u[np.arange(n), face] = np.sign(u[np.arange(n), face]) * 0.5 * np.array([b[5], b[4], b[3]])[face]
            c, s = math.cos(b[6]), math.sin(b[6])
            pts.append(np.stack([u[:, 0] * c - u[:, 1] * s + b[0], u[:, 0] * s + u[:, 1] * c + b[1], u[:, 2] + b[2]], 1))
        ground = np.stack([rs.uniform(rng[0], rng[3], n_ground), rs.uniform(rng[1], rng[4], n_ground), rs.uniform(-2.6, -2.3, n_ground)], 1)
        world = np.concatenate(pts + [ground]).astype(np.float64)
        Tinv = np.linalg.inv(poses[i].numpy())
        local = (Tinv[:3, :3] @ world.T).T + Tinv[:3, 3]
        clouds.append(np.concatenate([local, rs.uniform(0, 1, (len(local), 1))], 1).astype(np.float32))
    pw = O.pairwise_from_poses(poses, n_agents)[None].float()
    return clouds, pw, boxes, gt


# parameters of the trained-checkpoint fixture (tests/golden/ap_checkpoint.npz)
CKPT_SEED_WEIGHTS = 91          # model_fixture.model_state_dict seed of everything that is NOT trained
TRAIN_SCENE_SEED, EVAL_SCENE_SEED = 1001, 2002
TRAINED_PREFIXES = ("fusion_net.mlp_head.net.1.", "decoder.lidar_cls_head.", "decoder.lidar_reg_head.")
TRAINED_BN = ("decoder.lidar_decoder.decoder.1.", "decoder.lidar_decoder.decoder.4.", "decoder.lidar_decoder.decoder.7.",
              "decoder.lidar_decoder.decoder.10.")


def is_trained(key: str) -> bool:
    if key.startswith(TRAINED_PREFIXES):
        return True
    return key.startswith(TRAINED_BN) and key.endswith((".weight", ".bias"))
