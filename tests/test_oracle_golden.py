"""The CPU oracle (oracle/hmvit_oracle.py) replayed against golden vectors frozen from the
imported reference (tests/golden/make_goldens.py).  CPU-only; pins the oracle."""
import numpy as np
import os

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


@pytest.mark.skipif(not os.environ.get("HMVIT_SLOW"), reason="full-size oracle run: ~3 min and ~20 GB; set HMVIT_SLOW=1")
@pytest.mark.parametrize("name", ["g12_fusion_cfg2.npz", "g13_fusion_cfg3.npz", "g18_fusion_cfg4.npz"])
def test_full_size_goldens(name):
    """The oracle against the reference's forward at the headline size (BASELINE configs[1] / [2] / [3])."""
    g = load_golden(name)
    sd = O.random_state_dict(g["cfg"], g["seed_weights"])
    y = O.hetero_fusion(*O.synthetic_scene(**g["scene"]), sd, g["cfg"])
    scale = float(g["abs_max"])
    assert float((y[:, :, 3::8, 5::16] - g["out_sub"]).abs().max()) / scale < TOL
    assert float((y[:, :, [0, 199], :] - g["out_rows"]).abs().max()) / scale < TOL
    assert float((y.double().mean((0, 2, 3)) - g["chan_mean"]).abs().max()) < 1e-5


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


def _g10_frames(g):
    from oracle import postprocess_oracle as PPO
    params = PPO.make_params(W=96, H=64)
    anchors = PPO.generate_anchor_box(params)
    frames = []
    for seed0 in g["seeds"]:
        psm0, rm0, _, gt = PPO.synthetic_heads(params, seed=int(seed0), n_obj=14)
        psm1, rm1, _, _ = PPO.synthetic_heads(params, seed=int(seed0) + 1, n_obj=6)
        frames.append((psm0, rm0, psm1, rm1, gt))
    return params, anchors, frames


def test_g10_postprocess_and_ap():
    """post_process (decode, filters, rotated NMS, range mask) and the AP bookkeeping against the reference's own run
    (shapely's polygon intersection replaced by the oracle's clipper on both sides, see the module header)."""
    from oracle import postprocess_oracle as PPO
    g = load_golden("g10_postprocess.npz")
    params, anchors, frames = _g10_frames(g)
    stat = {t: {"tp": [], "fp": [], "gt": 0} for t in (0.3, 0.5, 0.7)}
    for k, (psm0, rm0, psm1, rm1, gt) in enumerate(frames):
        boxes, scores = PPO.post_process(params, [
            {"psm": psm0, "rm": rm0, "anchor_box": anchors, "transformation_matrix": np.eye(4, dtype=np.float32)},
            {"psm": psm1, "rm": rm1, "anchor_box": anchors, "transformation_matrix": g["T1"].numpy()}])
        ref_b, ref_s = g[f"boxes{k}"].numpy(), g[f"scores{k}"].numpy()
        assert boxes.shape == ref_b.shape
        assert np.abs(boxes - ref_b).max() < 1e-4 and np.abs(scores - ref_s).max() < 1e-6
        for t in stat:
            PPO.caluclate_tp_fp(boxes, scores, gt, stat, t)
    for t, tag in ((0.3, "30"), (0.5, "50"), (0.7, "70")):
        assert stat[t]["tp"] == g["tp" + tag].numpy().tolist() and stat[t]["fp"] == g["fp" + tag].numpy().tolist()
    ap = [PPO.calculate_ap(stat, t)[0] for t in (0.3, 0.5, 0.7)]
    assert np.allclose(ap, g["ap"].numpy(), atol=1e-12)


def test_g11_cross_view_attention():
    from oracle import cvt_oracle as CO
    g = load_golden("g11_cross_view.npz")
    for tag, no_feat, skip in (("a", False, True), ("b", True, False)):
        cfg = CO.make_config()
        cfg["no_image_features"], cfg["skip"] = no_feat, skip
        sd = CO.random_state_dict(64, 128, cfg, seed=int(g["seed_weights"]))
        x, feat, I_inv, E_inv = CO.synthetic_inputs(2, 4, 64, 12, 12, 128, 8, 8, seed=int(g["seed_inputs"]))
        y = CO.cross_view_attention(x, CO.bev_grid(64, 64, 100.0, 100.0, 0.0, 3), feat, I_inv, E_inv, sd, cfg)
        assert rel_max_err(y, g["y_" + tag]) < TOL


def test_g15_naive_compressor():
    from oracle import decoder_oracle as DO
    g = load_golden("g15_compressor.npz")
    x = torch.from_numpy(np.random.RandomState(int(g["seed_x"])).standard_normal((3, 256, 10, 12)).astype(np.float32))
    y = DO.naive_compressor(x, DO.compressor_state_dict(256, 4, seed=g["seed_weights"]))
    assert rel_max_err(y, g["out"]) < TOL


def test_g16_fax_modules():
    """CrossViewSwapAttention (both level kinds), Attention and the down-sampling block of the reference's FAX lift."""
    from oracle import fax_oracle as FO
    g = load_golden("g16_fax.npz")
    cfg = FO.make_swap_config(64)
    grids = FO.bev_grids(32, 32, 50.0, 50.0, 0.0, [2, 4])
    for index, (fh, H) in enumerate(((8, 16), (4, 8))):
        sd = FO.swap_state_dict(64, 128, cfg, index, seed=161 + index)
        x, feat, I_inv, E_inv = FO.synthetic_inputs(2, 3, 64, fh, fh, 128, H, H, seed=163 + index, image=64)
        y = FO.cross_view_swap_attention(x, grids[index], feat, I_inv, E_inv, sd, cfg, index)
        assert rel_max_err(y, g[f"swap{index}"]) < TOL, index
    asd = {k[len("attn_sd."):]: v for k, v in g.items() if k.startswith("attn_sd.")}
    assert rel_max_err(FO.self_attention(g["attn_x"], asd, 32, 8), g["self_attn"]) < TOL
    dsd = {f"downsample_layers.0.{k[len('down_sd.'):]}": v for k, v in g.items() if k.startswith("down_sd.")}
    assert rel_max_err(FO.downsample_block(g["down_x"], dsd, "downsample_layers.0"), g["down"]) < TOL
