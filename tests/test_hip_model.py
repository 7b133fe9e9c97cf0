"""End-to-end GPU parity: pillars -> PointPillar -> regroup -> HeteroFusion -> HeteroDecoder against the
reference model's own forward (golden g9, LiDAR-only batch with a ragged record_len)."""
import os
import sys

import pytest
import torch

from conftest import GOLDEN, load_golden, rel_max_err

sys.path.insert(0, GOLDEN)
pytestmark = pytest.mark.gpu


def _to(batch, dev):
    out = {}
    for k, v in batch.items():
        out[k] = _to(v, dev) if isinstance(v, dict) else v.to(dev)
    return out


@pytest.mark.parametrize("precision,tol", [("f32", 2e-4), ("f16", 3e-3)])
def test_model_matches_reference_forward(precision, tol):
    import hmvit_amd
    from model_fixture import model_batch, model_config, model_state_dict
    g = load_golden("g9_model.npz")
    cfg = model_config()
    net = hmvit_amd.BevformerPointPillarHetero(cfg, precision=precision)
    missing, unexpected = net.load_state_dict(model_state_dict(cfg, g["seed_weights"]), strict=False)
    assert not missing and not unexpected
    net = net.cuda().eval()
    batch = _to(model_batch(cfg, int(g["seed_batch"])), "cuda")
    keep = batch["processed_lidar"]["voxel_coords"].clone()
    out = net(batch)
    assert torch.equal(keep, batch["processed_lidar"]["voxel_coords"])          # inputs not mutated
    assert out["psm"].shape == g["psm"].shape and out["rm"].shape == g["rm"].shape
    assert rel_max_err(out["psm"].cpu(), g["psm"]) < tol
    assert rel_max_err(out["rm"].cpu(), g["rm"]) < tol


def test_camera_agents_need_a_camera_encoder():
    import hmvit_amd
    from model_fixture import model_batch, model_config
    cfg = model_config()
    net = hmvit_amd.BevformerPointPillarHetero(cfg).cuda().eval()
    batch = _to(model_batch(cfg, 95), "cuda")
    batch["mode"][0, 1] = 0
    with pytest.raises(NotImplementedError):
        net(batch)
