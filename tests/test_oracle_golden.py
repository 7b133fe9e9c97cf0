"""The CPU oracle (oracle/hmvit_oracle.py) replayed against golden vectors frozen from the
imported reference (tests/golden/make_goldens.py).  CPU-only; pins the oracle."""
import pytest
import torch

from conftest import load_golden, rel_max_err
from oracle import hmvit_oracle as O

TOL = 2e-5  # fp32 re-association only: the oracle performs the reference's operations


def test_g1_attention_intermediates():
    g = load_golden("g1_attention.npz")
    sd = O.random_state_dict(g["cfg"], g["seed_weights"])
    blk = g["cfg"]["hetero_fusion_block"]
    out, sim, attn = O.hetero_attention(g["xw"], g["mode"], g["mask"], sd,
                                        "hetero_fusion_block.window_attention",
                                        blk["dim_head"], blk["window_size"], True)
    B, X, Y, M, n, K = sim.shape
    sim_ref = g["sim"].reshape(B, X, Y, M, n, K)
    finite = torch.isfinite(sim_ref)
    assert torch.equal(finite, torch.isfinite(sim))
    assert float((sim[finite] - sim_ref[finite]).abs().max()) < 1e-5
    assert float((attn - g["attn"].reshape(attn.shape)).abs().max()) < 1e-6
    assert rel_max_err(out, g["out"]) < TOL


def test_g2_warp_and_roi():
    g = load_golden("g2_warp.npz")
    H, W = g["src"].shape[-2:]
    for c, (yaw, tx, ty) in enumerate(g["cases"].tolist()):
        T = O.rigid(yaw, tx, ty).to(torch.float32)[None, None]
        A = O.pixel_affine(T, 0.4, 4, H, W)[0]
        assert float((A[0] - g["A"][c]).abs().max()) < 1e-5
        y = O.warp_affine(g["src"], A)
        assert float((y[0] - g["bilinear"][c]).abs().max()) < 2e-5
        m = O.roi_and_cav_mask(H, W, torch.ones(1, 1), T, 0.4, 4)[0, :, :, 0, 0]
        assert int((m != g["roi"][c]).sum()) == 0


@pytest.mark.parametrize("name", ["g3_block_seq.npz", "g3_block_par.npz"])
def test_g3_block(name):
    g = load_golden(name)
    sd = O.random_state_dict(g["cfg"], g["seed_weights"])
    x, pw, mode, rl, mask = O.synthetic_scene(**g["scene"])
    y = O.hetero_fusion_block(x, pw, mode.long(), rl, mask, sd, "hetero_fusion_block",
                              g["cfg"]["hetero_fusion_block"])
    assert rel_max_err(y, g["out"]) < TOL


def test_g4_fusion_c256():
    g = load_golden("g4_fusion_c256.npz")
    sd = O.random_state_dict(g["cfg"], g["seed_weights"])
    y = O.hetero_fusion(*O.synthetic_scene(**g["scene"]), sd, g["cfg"])
    assert rel_max_err(y, g["out"]) < TOL


def test_g5_fusion_ragged_batch():
    g = load_golden("g5_fusion_ragged.npz")
    sd = O.random_state_dict(g["cfg"], g["seed_weights"])
    y = O.hetero_fusion(g["x"], g["pairwise"], g["mode"], g["record_len"], g["mask"], sd, g["cfg"])
    assert torch.isfinite(y).all()
    assert rel_max_err(y, g["out"]) < TOL


def test_g6_fusion_cfg1_full_size():
    g = load_golden("g6_fusion_cfg1.npz")
    sd = O.random_state_dict(g["cfg"], g["seed_weights"])
    y = O.hetero_fusion(*O.synthetic_scene(**g["scene"]), sd, g["cfg"])
    assert rel_max_err(y[:, :, ::5, ::11], g["out_sub"]) < TOL
    y64 = y.double()
    assert float((y64.mean((0, 2, 3)) - g["chan_mean"]).abs().max()) < 1e-5
    assert float((y64.abs().mean((0, 2, 3)) - g["chan_absmean"]).abs().max()) < 1e-5


def test_bad_architect_mode_raises():
    cfg = O.make_config(64, 4, 2, arch="bogus")
    sd = O.random_state_dict(O.make_config(64, 4, 2))
    with pytest.raises(ValueError):
        O.hetero_fusion(*O.synthetic_scene(2, 64, 8, 8, [1, 1]), sd, cfg)


def test_g7_pointpillar_encoder():
    from oracle import pointpillar_oracle as PO
    g = load_golden("g7_pointpillar.npz")
    nx, ny = [int(v) for v in g["grid"]]
    args = PO.make_args(nx, ny)
    sd = PO.random_state_dict(args, g["seed_weights"])
    vf, vc, vn = PO.synthetic_pillars(int(g["n_agents"]), int(g["n_per_agent"]), nx, ny, args, int(g["seed_pillars"]))
    pf = PO.pillar_vfe(vf, vn, vc, sd, args["voxel_size"], args["lidar_range"])
    assert rel_max_err(pf, g["pillar_features"]) < TOL
    y = PO.point_pillar_features(vf, vc, vn, sd, args, int(g["n_agents"]))
    assert y.shape == g["out"].shape
    assert rel_max_err(y, g["out"]) < TOL


def test_g8_hetero_decoder():
    import numpy as np
    from oracle import decoder_oracle as DO
    g = load_golden("g8_decoder.npz")
    params = DO.make_params()
    sd = DO.random_state_dict(params, g["seed_weights"])
    x = torch.from_numpy(np.random.RandomState(int(g["seed_x"])).standard_normal((3, 1, 256, 12, 10)).astype(np.float32))
    psm, rm = DO.hetero_decoder(x, g["mode"], sd, params)
    assert rel_max_err(psm, g["psm"]) < TOL and rel_max_err(rm, g["rm"]) < TOL
