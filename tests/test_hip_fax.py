"""HIP FAX camera lift (hm-vit_amd/fax.py) against the reference's own forward for the attention modules (golden
g16: CrossViewSwapAttention of both level kinds, the closing self-attention) and against the CPU restatement
(oracle/fax_oracle.py + camera_oracle.py) for the assembled encoder in the HM-ViT camera slot."""
import pytest
import torch

from conftest import load_golden, rel_max_err
from oracle import camera_oracle as CAM
from oracle import fax_oracle as FO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("f16", 1e-2)])
@pytest.mark.parametrize("index", [0, 1])
def test_cross_view_swap_attention_matches_reference_golden(index, precision, tol):
    from hmvit_amd.fax import BEVEmbedding, CrossViewSwapAttention
    g = load_golden("g16_fax.npz")
    cfg = FO.make_swap_config(64)
    fh, H = ((8, 16), (4, 8))[index]
    net = CrossViewSwapAttention(fh, fh, 64, 128, index, **{k: cfg[k] for k in (
        "image_height", "image_width", "no_image_features", "skip", "heads", "dim_head", "qkv_bias", "rel_pos_emb", "q_win_size",
        "feat_win_size", "bev_embedding_flag")})
    sd = FO.swap_state_dict(64, 128, cfg, index, seed=161 + index)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    net = net.cuda().eval()
    net.precision = precision
    bev = BEVEmbedding(128, 1.0, 32, 32, 50.0, 50.0, 0.0, [2, 4]).cuda()
    x, feat, I_inv, E_inv = FO.synthetic_inputs(2, 3, 64, fh, fh, 128, H, H, seed=163 + index, image=64)
    y = net(index, x.cuda(), bev, feat.cuda(), I_inv.cuda(), E_inv.cuda()).cpu()
    assert rel_max_err(y, g[f"swap{index}"]) < tol


def test_self_attention_with_relative_position_bias_matches_reference_golden():
    from hmvit_amd.fax import Attention
    g = load_golden("g16_fax.npz")
    net = Attention(128, dim_head=32, dropout=0.1, window_size=8)
    net.load_state_dict({k[len("attn_sd."):]: v for k, v in g.items() if k.startswith("attn_sd.")}, strict=True)
    net = net.cuda().eval()
    assert rel_max_err(net(g["attn_x"].cuda()).cpu(), g["self_attn"]) < 1e-4


def _camera_net(cfg, precision, seed):
    import hmvit_amd
    torch.manual_seed(seed)
    net = hmvit_amd.FaxCameraEncoder(cfg, precision=precision)
    with torch.no_grad():                                   # non-trivial BatchNorm statistics / LayerNorm affines
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.6, 1.4); m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
    net.set_return_features()
    return net.cuda().eval()


@pytest.mark.parametrize("precision,tol", [("f32", 3e-4), ("split", 3e-4), ("f16", 1e-2)])
def test_fax_camera_encoder_vs_oracle(precision, tol):
    cfg = FO.make_camera_config(image=64)
    net = _camera_net(cfg, precision, seed=9)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    batch = CAM.synthetic_batch(2, CAM.make_config(image=64), seed=10)
    ref = FO.fax_camera_encoder(batch, sd, cfg)
    y = net({k: v.cuda() for k, v in batch.items()}).cpu()
    assert y.shape == ref.shape == (2, 256, 32, 32)
    assert rel_max_err(y, ref) < tol


def test_fax_encoder_fills_the_model_camera_slot():
    """BevformerPointPillarHetero with FaxCameraEncoder as `camera_encoder` (the slot BEVFormer fills in the reference,
    bevformer_point_pillar_hetero.py:55,106-108): camera ego + LiDAR collaborator, against the CPU composition."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import hmvit_amd
    from model_fixture import model_config, model_state_dict
    from oracle import decoder_oracle as DO, hmvit_oracle as O, pointpillar_oracle as PO
    ccfg = FO.make_camera_config(image=64)
    cam = _camera_net(ccfg, "f32", seed=11)
    cfg = model_config()
    # the LiDAR branch must produce the camera branch's map size: 32 x 32 BEV <- 128 x 128 pillar canvas
    cfg["lidar"] = PO.make_args(128, 128)
    cfg["hetero_fusion"] = O.make_config(256, 4, 2, voxel=0.4, downsample=4)
    cfg["max_cav"] = 2
    sd = model_state_dict(cfg, 71)
    net = hmvit_amd.BevformerPointPillarHetero(cfg, camera_encoder=cam, precision="f32")
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    vf, vc, vn = PO.synthetic_pillars(1, 300, 128, 128, cfg["lidar"], seed=72)
    batch_cam = CAM.synthetic_batch(1, CAM.make_config(image=64), seed=73)
    _, pw, _, _, _ = O.synthetic_scene(2, 1, 1, 1, [0, 1], seed=0, tx_step=3.0, ty_step=-2.0)
    mode = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
    batch = {"mode": mode.cuda(), "record_len": torch.tensor([2]).cuda(), "pairwise_t_matrix": pw.cuda(),
             "processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": torch.cat([torch.ones(len(vc), 1, dtype=vc.dtype), vc[:, 1:]], 1).cuda(),
                                 "voxel_num_points": vn.cuda()},
             "camera": batch_cam["camera"].cuda(), "intrinsic": batch_cam["intrinsic"].cuda(), "extrinsic": batch_cam["extrinsic"].cuda(),
             "cav2cam_extrinsic": batch_cam["extrinsic"].cuda()}
    # the batch's per-agent camera tensors are indexed by the flat agent index: agent 0 camera, agent 1 LiDAR
    for k in ("camera", "intrinsic", "extrinsic", "cav2cam_extrinsic"):
        batch[k] = torch.cat([batch[k], torch.zeros_like(batch[k])], 0)
    with torch.no_grad():
        out = net(batch)
    csd = {k: v.detach().cpu() for k, v in cam.state_dict().items()}
    f_cam = FO.fax_camera_encoder(batch_cam, csd, ccfg)
    lsd = {k[len("lidar_encoder."):]: v for k, v in sd.items() if k.startswith("lidar_encoder.")}
    f_lid = PO.point_pillar_features(vf, vc, vn, lsd, cfg["lidar"], 1)
    x = torch.cat([f_cam, f_lid], 0)[None]
    fsd = {k[len("fusion_net."):]: v for k, v in sd.items() if k.startswith("fusion_net.")}
    fused = O.hetero_fusion(x, pw, mode.int(), torch.tensor([2]), torch.ones(1, 2, dtype=torch.int64), fsd, cfg["hetero_fusion"])
    psm, rm = DO.hetero_decoder(fused.unsqueeze(1), mode.int(), {k: v for k, v in sd.items() if k.startswith("decoder.")},
                                cfg["hetero_decoder"], prefix="decoder")
    assert rel_max_err(out["psm"].cpu(), psm) < 1e-3 and rel_max_err(out["rm"].cpu(), rm) < 1e-3
