"""Seeded inputs shared by make_goldens.py (build container, imports the reference) and the tests (no reference)."""
import numpy as np
import torch


def loss_inputs(seed=141):
    """Seeded head outputs and anchor targets for the loss fixture (shared with tests/test_train_cpu.py)."""
    rs = np.random.RandomState(seed)
    B, A, H, W = 2, 2, 8, 12
    psm = torch.from_numpy(rs.standard_normal((B, A, H, W)).astype(np.float32))
    rm = torch.from_numpy(rs.standard_normal((B, 7 * A, H, W)).astype(np.float32))
    targets = torch.from_numpy(rs.standard_normal((B, H, W, 7 * A)).astype(np.float32))
    pos = torch.from_numpy((rs.uniform(size=(B, H, W, A)) > 0.93).astype(np.float32))
    pos[1] = 0                       # a sample without positives: the normaliser clamps at 1
    targets[0, 2, 3, 4] = float("nan")
    return psm, rm, {"targets": targets, "pos_equal_one": pos}


def label_inputs(seed=171, n_obj=9, max_num=16):
    """Seeded ground-truth boxes for the anchor-target fixture g17 (hwl order, a few near each other so that anchors are
    claimed by more than one box; one mask hole at the end as the padding of ``object_bbx_center`` produces)."""
    rs = np.random.RandomState(seed)
    gt = np.zeros((max_num, 7), np.float32)
    mask = np.zeros(max_num, np.float32)
    for i in range(n_obj):
        near = i > 0 and i % 3 == 0
        cx = gt[i - 1, 0] + rs.uniform(-1.5, 1.5) if near else rs.uniform(-30, 30)
        cy = gt[i - 1, 1] + rs.uniform(-1.5, 1.5) if near else rs.uniform(-18, 18)
        gt[i] = [cx, cy, rs.uniform(-1.4, -0.6), rs.uniform(1.4, 1.8), rs.uniform(1.5, 2.1), rs.uniform(3.5, 4.8),
                 rs.uniform(-np.pi, np.pi)]
        mask[i] = 1
    return gt, mask


def label_params(W=96, H=64):
    """postprocess block of the shipped yaml (opcl/bevformer_point_pillar_hetero.yaml:56-76) on a small grid."""
    return {"core_method": "VoxelPostprocessor", "order": "hwl", "max_num": 100, "nms_thresh": 0.15,
            "anchor_args": {"cav_lidar_range": [-38.4, -25.6, -3, 38.4, 25.6, 1], "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90],
                            "num": 2, "feature_stride": 2, "vw": 0.4, "vh": 0.4, "vd": 4, "W": W, "H": H, "D": 1},
            "target_args": {"pos_threshold": 0.6, "neg_threshold": 0.45, "score_threshold": 0.27}}
