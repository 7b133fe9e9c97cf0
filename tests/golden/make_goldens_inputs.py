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
