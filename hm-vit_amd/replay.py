"""Inference replay on procedurally generated scenes: the per-frame loop of the reference's inference driver
(``opencood/tools/inference_camera.py:145-195``: batch -> model -> ``post_process`` -> ``caluclate_tp_fp`` at IoU 0.3 / 0.5 /
0.7 -> ``calculate_ap``) over a synthetic stand-in for the OPV2V dataset, all of it on the HIP pipeline (pillariser ->
PointPillar -> HeteroFusion -> HeteroDecoder -> heads -> box decode / rotated NMS -> AP arithmetic).

There is no OPV2V data and no trained checkpoint in this environment, so a scene is: vehicles of the anchor size
scattered over the ego's range, every agent (rigid poses T_i, ``pairwise[i, j] = inv(T_j) T_i`` as
``mixed/intermediate_fusion_dataset.py:163-202`` builds them) seeing points on the vehicle surfaces plus ground clutter in
its own frame.  The config dicts carry the keys of the shipped yaml (``opencood/hypes_yaml/opcamera/mixed/hm_vit.yaml``).
With random weights the AP is that of an untrained detector; what the harness provides is the loop, its timing and a place
to load a reference checkpoint (``--checkpoint``: a ``state_dict`` file, loaded ``strict=False`` as ``train_utils.py:66-70``).

    python -m hmvit_amd.replay --scenes 8 --precision f16        (from the repository root)
"""
from __future__ import annotations

import argparse
import json
import math
import time

import numpy as np
import torch

from . import synthetic as S
from .postprocess import boxes_to_corners_3d


def lidar_model_config(nx: int = 512, ny: int = 512, max_cav: int = 5, window: int = 8, small: bool = False) -> dict:
    """LiDAR-only HM-ViT model config with the shipped yaml's structure: 0.4 m pillars on an (nx, ny) grid, PointPillar with
    layer_nums [3, 5, 8] (``small``: [1, 2, 2]), shrink header to 256 channels at 1/4 resolution, fusion with C = 256."""
    lidar = {"voxel_size": [0.4, 0.4, 4], "lidar_range": [-nx * 0.2, -ny * 0.2, -3, nx * 0.2, ny * 0.2, 1],
             "anchor_number": 2, "cls_head_dim": 256,
             "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
             "point_pillar_scatter": {"num_features": 64, "grid_size": [nx, ny, 1]},
             "base_bev_backbone": {"layer_nums": [1, 2, 2] if small else [3, 5, 8], "layer_strides": [2, 2, 2],
                                   "num_filters": [64, 128, 256], "upsample_strides": [1, 2, 4],
                                   "num_upsample_filter": [128, 128, 128]},
             "shrink_header": {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [256], "input_dim": 384}}
    st = {"downsample_rate": 4, "voxel_size": [0.4, 0.4, 4], "use_roi_mask": True}
    return {"max_cav": max_cav, "anchor_number": 2, "compression": 0, "spatial_transform": st, "camera": {}, "lidar": lidar,
            "hetero_fusion": S.make_config(256, window, max_cav, voxel=0.4, downsample=4),
            "hetero_decoder": {"input_dim": 256, "num_layer": 2, "num_ch_dec": [256, 256], "anchor_number": 2}}


def preprocess_params(cfg: dict) -> dict:
    la = cfg["lidar"]
    return {"cav_lidar_range": la["lidar_range"],
            "args": {"voxel_size": la["voxel_size"], "max_points_per_voxel": 32, "max_voxel_train": 32000, "max_voxel_test": 70000}}


def postprocess_params(cfg: dict) -> dict:
    la = cfg["lidar"]
    nx, ny = la["point_pillar_scatter"]["grid_size"][:2]
    return {"order": "hwl", "nms_thresh": 0.15, "max_num": 100,
            "target_args": {"score_threshold": 0.27, "pos_threshold": 0.6, "neg_threshold": 0.45},
            "anchor_args": {"cav_lidar_range": la["lidar_range"], "W": nx // 2, "H": ny // 2, "vw": 0.4, "vh": 0.4, "vd": 4,
                            "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90], "num": 2, "feature_stride": 2}}


class SyntheticReplayDataset:
    """``len(ds)`` frames; ``ds[i]`` -> {'clouds': [(n_i, 4) float32 per agent], 'pairwise_t_matrix' (1, L, L, 4, 4),
    'mode' (1, L), 'record_len' (1,), 'object_bbx_corners' (n_obj, 8, 3)} - what the driver needs from ``batch['ego']``."""

    def __init__(self, cfg: dict, n_frames: int, n_agents: int | None = None, n_obj: int = 12, seed: int = 7):
        self.cfg, self.n_frames, self.n_obj = cfg, n_frames, n_obj
        self.n_agents = n_agents or cfg["max_cav"]
        self.seed = seed

    def __len__(self):
        return self.n_frames

    def __getitem__(self, idx):
        rs = np.random.RandomState(self.seed + 1000 * idx)
        rng = self.cfg["lidar"]["lidar_range"]
        L = self.n_agents
        span = min(rng[3], rng[4])
        poses = [S.rigid(0.15 * i, 0.06 * span * i, -0.04 * span * i) for i in range(L)]        # agent i -> ego frame
        boxes = np.array([[rs.uniform(rng[0] + 3, rng[3] - 3), rs.uniform(rng[1] + 3, rng[4] - 3), -1.0, 1.56, 1.6, 3.9,
                           rs.uniform(-math.pi, math.pi)] for _ in range(self.n_obj)], np.float32)
        clouds = []
        for i in range(L):
            pts = []
            for b in boxes:
                dims = np.array([b[5], b[4], b[3]])
                u = rs.uniform(-0.5, 0.5, (150, 3)) * dims
                face = rs.randint(0, 3, 150)
                u[np.arange(150), face] = np.sign(u[np.arange(150), face]) * 0.5 * dims[face]
                c, s = math.cos(b[6]), math.sin(b[6])
                pts.append(np.stack([u[:, 0] * c - u[:, 1] * s + b[0], u[:, 0] * s + u[:, 1] * c + b[1], u[:, 2] + b[2]], 1))
            n_ground = 40 * int(rng[3] - rng[0])
            ground = np.stack([rs.uniform(rng[0], rng[3], n_ground), rs.uniform(rng[1], rng[4], n_ground),
                               rs.uniform(-2.6, -2.3, n_ground)], 1)
            world = np.concatenate(pts + [ground]).astype(np.float64)
            Tinv = np.linalg.inv(poses[i].numpy())
            local = (Tinv[:3, :3] @ world.T).T + Tinv[:3, 3]
            clouds.append(np.concatenate([local, rs.uniform(0, 1, (len(local), 1))], 1).astype(np.float32))
        return {"clouds": clouds, "pairwise_t_matrix": S.pairwise_from_poses(poses, self.cfg["max_cav"])[None],
                "mode": torch.ones(1, self.cfg["max_cav"], dtype=torch.float64), "record_len": torch.tensor([L]),
                "object_bbx_corners": boxes_to_corners_3d(boxes), "object_bbx_center_valid": boxes}


def inference(model, dataset, pre, post, calibrate_top: int | None = 200, log=None) -> dict:
    """The driver loop.  ``calibrate_top``: with an untrained head every anchor fires, so the score threshold of each frame
    is set to keep that many anchors (None: the config's threshold, for a trained checkpoint)."""
    from .postprocess import calculate_ap, caluclate_tp_fp
    dev = next(model.parameters()).device
    anchors = post.generate_anchor_box()
    thr = (0.3, 0.5, 0.7)
    stat = {t: {"tp": [], "fp": [], "gt": 0} for t in thr}
    n_det, t_model, t_post = 0, 0.0, 0.0
    for i in range(len(dataset)):
        frame = dataset[i]
        lidar = pre.collate_batch([pre.preprocess(c) for c in frame["clouds"]])
        batch = {"mode": frame["mode"].to(dev), "record_len": frame["record_len"].to(dev),
                 "pairwise_t_matrix": frame["pairwise_t_matrix"].to(dev), "processed_lidar": lidar}
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = model(batch)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        if out["psm"].shape[2:] != anchors.shape[:2]:
            raise RuntimeError(f"anchor grid {anchors.shape[:2]} != head grid {tuple(out['psm'].shape[2:])}")
        if calibrate_top is not None:
            k = max(1, out["psm"].numel() - calibrate_top)
            kth = float(torch.sigmoid(out["psm"].float()).flatten().kthvalue(k).values)
            post.params["target_args"]["score_threshold"] = float(np.nextafter(np.float32(min(kth, 1.0)), np.float32(0)))
        data = {"ego": {"anchor_box": torch.from_numpy(anchors), "transformation_matrix": torch.eye(4)}}
        boxes, scores = post.post_process(data, {"ego": {"psm": out["psm"], "rm": out["rm"]}})
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        gt = torch.from_numpy(frame["object_bbx_corners"]).to(dev)
        for t in thr:
            caluclate_tp_fp(boxes, scores, gt, stat, t)
        n_det += 0 if boxes is None else len(boxes)
        if i > 0:                      # the first frame carries the weight preparation
            t_model += t1 - t0
            t_post += t2 - t1
        if log:
            log(f"frame {i}: {0 if boxes is None else len(boxes)} boxes, model {1e3 * (t1 - t0):.2f} ms, post {1e3 * (t2 - t1):.2f} ms")
    n = max(1, len(dataset) - 1)
    res = {"frames": len(dataset), "detections": n_det, "model_ms_per_frame": 1e3 * t_model / n,
           "postprocess_ms_per_frame": 1e3 * t_post / n}
    for t in thr:
        res[f"AP@{t}"] = round(100 * calculate_ap(stat, t)[0], 3) if stat[t]["tp"] else 0.0
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--grid", type=int, nargs=2, default=[512, 192], metavar=("NX", "NY"),
                    help="pillar grid (0.4 m cells); boxes outside the evaluation range |y| <= 40 m (GT_RANGE) are dropped "
                         "by the post-processor as in the reference, so the default keeps the map inside it")
    ap.add_argument("--precision", default="f16", choices=["f16", "f32"])
    ap.add_argument("--checkpoint", default=None, help="state_dict file of the reference model (loaded strict=False)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args(argv)
    from . import BevformerPointPillarHetero, SpVoxelPreprocessor, VoxelPostprocessor
    if not torch.cuda.is_available():
        raise SystemExit("hm-vit_amd has no CPU path: this needs an MI355X")
    cfg = lidar_model_config(args.grid[0], args.grid[1], max_cav=args.agents)
    torch.manual_seed(args.seed)
    model = BevformerPointPillarHetero(cfg, precision=args.precision)
    if args.checkpoint:
        missing, unexpected = model.load_state_dict(torch.load(args.checkpoint, map_location="cpu"), strict=False)
        print(f"checkpoint: {len(missing)} missing / {len(unexpected)} unexpected keys")
    else:
        # untrained weights: keep the regression deltas small so that the decoded boxes stay anchor-like and survive the
        # post-processor's size filters (a random head otherwise decodes to boxes hundreds of metres long)
        with torch.no_grad():
            model.reg_head.weight.mul_(0.02)
            model.reg_head.bias.zero_()
            model.cls_head.weight.mul_(0.05)       # and the logits out of the sigmoid's saturated range (distinct scores)
    model = model.cuda().eval()
    pre = SpVoxelPreprocessor(preprocess_params(cfg), train=False)
    post = VoxelPostprocessor(postprocess_params(cfg), train=False)
    ds = SyntheticReplayDataset(cfg, args.scenes, n_agents=args.agents, seed=args.seed + 7)
    res = inference(model, ds, pre, post, calibrate_top=None if args.checkpoint else 200, log=print if args.verbose else None)
    res.update(precision=args.precision, agents=args.agents, grid=args.grid)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
