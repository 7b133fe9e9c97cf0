"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); nothing here travels to the GPU
box except the .npz files it writes.  Usage:  python tests/golden/make_goldens.py

What is frozen:
  g1_attention.npz      HeteroAttention.forward, L=3 mixed types, C=64, 2x3 windows of 4x4,
                        random key mask with the ego column forced to 1; stores sim (pre
                        softmax), attn and the output.
  g2_warp.npz           warp_affine (bilinear) + get_roi_and_cav_mask for several yaw /
                        translation pairs on 32x48 maps.
  g3_block_seq.npz /    HeteroFusionBlock sequential / parallel, L=3, record_len=2 (one
  g3_block_par.npz      padded agent), C=64, 16x24, window 4.
  g4_fusion_c256.npz    HeteroFusion, 2 iters, L=5 modes 10110, C=256, 16x16, window 8.
  g5_fusion_ragged.npz  HeteroFusion, B=2, record_len=[3,2], C=64, 16x16, window 4.
  g6_fusion_cfg1.npz    HeteroFusion at BASELINE configs[0]: 2 LiDAR agents, 100x352, C=64,
                        window 4 -- output sub-sampled (every 5th row / 11th col) + moments.
  g12_fusion_cfg2.npz   HeteroFusion at BASELINE configs[1] FULL SIZE: 5 LiDAR agents (modes 11111), 200x704, C=256,
                        window 8, 2 iters, 0.4 m/px, poses of SURVEY 8(d) -- output sub-sampled (every 8th row + row
                        offset 3 / every 16th col + col offset 5) + the first and last full rows + per-channel moments.
                        ~2.5 min and ~16 GB on 8 cores.
  g13_fusion_cfg3.npz   same at configs[2]'s type pattern 10110 (mixed camera / LiDAR agent types).
  g18_fusion_cfg4.npz   same at configs[3]'s type pattern 00000 (all camera agents, ego_mode=camera).
  g14_loss.npz          PointPillarLoss.forward (loss/point_pillar_loss.py:68-142; cls_weight 1, reg 2) on seeded head outputs /
                        targets (B=2, 2 anchors, 8x12, a few positives, one NaN target): total / reg / conf loss and the
                        gradients with respect to psm and rm.
  g15_compressor.npz    NaiveCompressor.forward (naive_compress.py:5-28, eval BatchNorm), input_dim 256, ratio 4, 3 maps of 10 x 12.
  g16_fax.npz           FAX lift (fax_modules.py, torchvision stubbed): CrossViewSwapAttention.forward for level 0 (BEV embedding,
                        16x16 queries in 8x8 windows, 8x8 features in 4x4 windows) and level 1 (no BEV embedding, 8x8 queries in
                        4x4 windows, 4x4 features in 2x2 windows), 2 agents x 3 cameras, dim 128; Attention.forward (self
                        attention with relative-position bias, window 8); one down-sampling block of FAXModule.
  g7_pointpillar.npz    PointPillar.forward (features only, eval BN): 2 agents x 400 pillars on a
                        64x48 canvas; PFN output + (2, 256, 12, 16) BEV features.
  g8_decoder.npz        HeteroDecoder.forward (no upsample), 3 samples with ego types 1,0,1, 12x10.
  g9_model.npz          BevformerPointPillarHetero.forward, LiDAR-only batch B=2, record_len [3,2]:
                        pillars -> PointPillar -> regroup -> HeteroFusion -> HeteroDecoder -> psm / rm.
  g11_cross_view.npz    CrossViewAttention.forward (cvt_modules.py:176-280, eval BatchNorm) with the grid of a BEVEmbedding:
                        2 agents x 4 cameras, feature maps (64, 12, 12), BEV queries 8 x 8, dim 128, 4 heads; also the
                        no_image_features / no-skip variant.  torchvision (absent) is stubbed: Bottleneck is not used here.
  g10_postprocess.npz   VoxelPostprocessor.post_process (2 agents, one projected by a rigid transform) on seeded head
                        outputs over a 32x48x2 anchor grid, and eval_utils.caluclate_tp_fp / calculate_ap over two
                        frames at IoU 0.3 / 0.5 / 0.7.  shapely is absent: its Polygon is replaced by the convex
                        clipper of oracle/postprocess_oracle.py, so everything but the polygon intersection is the
                        reference's own arithmetic.

Weights and inputs are NOT stored where they can be regenerated bit-exactly from a numpy
legacy RandomState seed (oracle.hmvit_oracle.random_state_dict / synthetic_scene); the
files then hold the seeds, the config and the reference's outputs.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


_stub("shapely")
_stub("shapely.geometry", Polygon=object)                       # common_utils.py:8 (NMS only)
_stub("opencood.models.bevformer_wrapper", BEVFormerWrapper=object)  # mmdet3d, out of scope

from opencood.models.bevformer_point_pillar_hetero import HeteroFusion  # noqa: E402
from opencood.models.sub_modules.hetero_fusion import HeteroAttention, HeteroFusionBlock  # noqa: E402
from opencood.models.sub_modules.torch_transformation_utils import (  # noqa: E402
    get_discretized_transformation_matrix, get_roi_and_cav_mask, get_transformation_matrix,
    warp_affine)

from oracle import hmvit_oracle as O  # noqa: E402  (only for the seeded input generators)
from oracle import pointpillar_oracle as PO  # noqa: E402  (seeded pillars / weights)
from oracle import decoder_oracle as DO  # noqa: E402  (seeded weights)
from model_fixture import model_batch, model_config, model_state_dict  # noqa: E402
from make_goldens_inputs import loss_inputs  # noqa: E402

torch.set_grad_enabled(False)


def load_into(module, sd, prefix=""):
    own = module.state_dict()
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing = [k for k in own if k not in sub]
    assert not missing, missing
    module.load_state_dict({k: sub[k] for k in own}, strict=True)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.numpy()
        if isinstance(v, (dict, list)):
            v = np.frombuffer(json.dumps(v).encode(), dtype=np.uint8)
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def g1_attention():
    C, w, L, X, Y, dh = 64, 4, 3, 2, 3, 32
    cfg = O.make_config(C, w, L)
    sd = O.random_state_dict(cfg, seed=11)
    att = HeteroAttention(C, dh, 0.1, L, w).eval()
    load_into(att, sd, "hetero_fusion_block.window_attention.")
    rs = np.random.RandomState(12)
    xw = torch.from_numpy(rs.standard_normal((1, L, X, Y, w, w, C)).astype(np.float32))
    mask = torch.from_numpy((rs.uniform(size=(1, X, Y, w, w, 1, L)) > 0.35).astype(np.float32))
    mask[..., 0] = 1.0
    mode = torch.tensor([[1, 0, 1]], dtype=torch.int32)
    captured = {}
    orig = att.attend

    class Tap(torch.nn.Module):
        def forward(self, s):
            captured["sim"] = s.clone()
            a = orig(s)
            captured["attn"] = a.clone()
            return a

    att.attend = Tap()
    out = att(xw, mode, mask=mask)
    save("g1_attention.npz", cfg=cfg, seed_weights=11, xw=xw, mask=mask, mode=mode,
         sim=captured["sim"], attn=captured["attn"], out=out)


def g2_warp():
    H, W, C = 32, 48, 3
    rs = np.random.RandomState(21)
    src = torch.from_numpy(rs.standard_normal((1, C, H, W)).astype(np.float32))
    cases = [(0.0, 0.0, 0.0), (0.0, 3.2, -1.6), (0.0, 1.3, 0.7), (0.3, 4.0, -2.5),
             (np.pi / 2, 2.0, 1.0), (-1.1, -6.3, 3.9), (3.0, 0.4, 0.4), (0.05, 40.0, 0.0)]
    bil, near, mats = [], [], []
    for yaw, tx, ty in cases:
        T = O.rigid(yaw, tx, ty).to(torch.float32)[None, None]
        disc = get_discretized_transformation_matrix(T, 0.4, 4)
        A = get_transformation_matrix(disc.reshape(-1, 2, 3), (H, W))
        mats.append(A[0].clone())
        bil.append(warp_affine(src, A, (H, W))[0])
        m = get_roi_and_cav_mask((1, 1, H, W, C), torch.ones(1, 1), T, 0.4, 4)
        near.append(m[0, :, :, 0, 0])
    save("g2_warp.npz", src=src, cases=np.array(cases, dtype=np.float64),
         A=torch.stack(mats), bilinear=torch.stack(bil), roi=torch.stack(near))


def g3_block(arch):
    C, w, L, H, W = 64, 4, 3, 16, 24
    cfg = O.make_config(C, w, L, arch=arch)
    sd = O.random_state_dict(cfg, seed=31)
    blk = HeteroFusionBlock(cfg["hetero_fusion_block"]).eval()
    load_into(blk, sd, "hetero_fusion_block.")
    x, pw, mode, rl, mask = O.synthetic_scene(L, C, H, W, [0, 1, 0], n_valid=2, seed=32,
                                              tx_step=6.0, ty_step=-4.0)
    y = blk(x, pw, mode, rl, mask)
    save(f"g3_block_{'seq' if arch == 'sequential' else 'par'}.npz", cfg=cfg, seed_weights=31,
         scene=dict(L=L, C=C, H=H, W=W, modes=[0, 1, 0], n_valid=2, seed=32, tx_step=6.0,
                    ty_step=-4.0), out=y)


def _run_fusion(cfg, sd, scene_kw, B=1):
    net = HeteroFusion(cfg).eval()
    load_into(net, sd)
    x, pw, mode, rl, mask = O.synthetic_scene(B=B, **scene_kw)
    return net, (x, pw, mode, rl, mask)


def g4_fusion_c256():
    cfg = O.make_config(256, 8, 5)
    sd = O.random_state_dict(cfg, seed=41)
    kw = dict(L=5, C=256, H=16, W=16, modes=[1, 0, 1, 1, 0], seed=42, tx_step=4.0, ty_step=-2.4)
    net, (x, pw, mode, rl, mask) = _run_fusion(cfg, sd, kw)
    y = net(x, pw, mode, rl, mask)
    save("g4_fusion_c256.npz", cfg=cfg, seed_weights=41, scene=kw, out=y)


def g5_fusion_ragged():
    cfg = O.make_config(64, 4, 3)
    sd = O.random_state_dict(cfg, seed=51)
    kw = dict(L=3, C=64, H=16, W=16, modes=[1, 0, 0], seed=52, tx_step=5.0, ty_step=3.0)
    net, (x, pw, mode, rl, mask) = _run_fusion(cfg, sd, kw, B=2)
    # sample 1 has only 2 agents: zero its third map, identity transforms, mode/mask 0
    x[1, 2] = 0
    eye = torch.eye(4)
    pw[1, 2, :] = eye
    pw[1, :, 2] = eye
    mode[1, 2] = 0
    mask[1, 2] = 0
    rl = torch.tensor([3, 2])
    y = net(x, pw, mode, rl, mask)
    save("g5_fusion_ragged.npz", cfg=cfg, seed_weights=51, scene=kw, x=x, pairwise=pw,
         mode=mode, record_len=rl, mask=mask, out=y)


def g6_fusion_cfg1():
    cfg = O.make_config(64, 4, 2, voxel=0.4, downsample=2)
    sd = O.random_state_dict(cfg, seed=61)
    kw = dict(L=2, C=64, H=100, W=352, modes=[1, 1], seed=1)
    net, (x, pw, mode, rl, mask) = _run_fusion(cfg, sd, kw)
    y = net(x, pw, mode, rl, mask)
    y64 = y.double()
    save("g6_fusion_cfg1.npz", cfg=cfg, seed_weights=61, scene=kw,
         out_sub=y[:, :, ::5, ::11].contiguous(),
         chan_mean=y64.mean((0, 2, 3)), chan_absmean=y64.abs().mean((0, 2, 3)),
         abs_max=y64.abs().max())


def _fusion_full_size(name, modes, seed_w, seed_x):
    """HeteroFusion of the reference at the headline size (BASELINE configs[1] / [2])."""
    import time
    cfg = O.make_config(256, 8, 5, voxel=0.4, downsample=1)
    sd = O.random_state_dict(cfg, seed=seed_w)
    kw = dict(L=5, C=256, H=200, W=704, modes=modes, seed=seed_x)
    net, (x, pw, mode, rl, mask) = _run_fusion(cfg, sd, kw)
    t0 = time.time()
    y = net(x, pw, mode, rl, mask)
    print(f"{name}: reference forward {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
    y64 = y.double()
    save(name, cfg=cfg, seed_weights=seed_w, scene=kw,
         out_sub=y[:, :, 3::8, 5::16].contiguous(), out_rows=y[:, :, [0, 199], :].contiguous(),
         chan_mean=y64.mean((0, 2, 3)), chan_absmean=y64.abs().mean((0, 2, 3)),
         chan_sqmean=(y64 * y64).mean((0, 2, 3)), abs_max=y64.abs().max())


def g12_fusion_cfg2():
    _fusion_full_size("g12_fusion_cfg2.npz", [1, 1, 1, 1, 1], 121, 1)


def g13_fusion_cfg3():
    _fusion_full_size("g13_fusion_cfg3.npz", [1, 0, 1, 1, 0], 131, 2)


def g18_fusion_cfg4():
    _fusion_full_size("g18_fusion_cfg4.npz", [0, 0, 0, 0, 0], 181, 3)


def g16_fax():
    from oracle import fax_oracle as FO
    _stub("torchvision"); _stub("torchvision.models"); _stub("torchvision.models.resnet", Bottleneck=object)
    from opencood.models.sub_modules.fax_modules import Attention, BEVEmbedding, CrossViewSwapAttention
    cfg = FO.make_swap_config(64)
    be = dict(sigma=1.0, bev_height=32, bev_width=32, h_meters=50.0, w_meters=50.0, offset=0.0, upsample_scales=[2, 4])
    bev = BEVEmbedding(128, **be)
    grids = FO.bev_grids(be["bev_height"], be["bev_width"], be["h_meters"], be["w_meters"], be["offset"], be["upsample_scales"])
    assert torch.allclose(bev.grid0, grids[0]) and torch.allclose(bev.grid1, grids[1])
    out = {}
    cv = {k: cfg[k] for k in ("image_height", "image_width", "no_image_features", "skip", "heads", "dim_head", "qkv_bias")}
    sw = {k: cfg[k] for k in ("rel_pos_emb", "q_win_size", "feat_win_size", "bev_embedding_flag")}
    for index, (fh, H) in enumerate(((8, 16), (4, 8))):
        net = CrossViewSwapAttention(fh, fh, 64, 128, index, **cv, **sw).eval()
        sd = FO.swap_state_dict(64, 128, cfg, index, seed=161 + index)
        own = net.state_dict()
        missing = [k for k in own if k not in sd and "num_batches_tracked" not in k]
        assert not missing, missing
        net.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        x, feat, I_inv, E_inv = FO.synthetic_inputs(2, 3, 64, fh, fh, 128, H, H, seed=163 + index, image=64)
        out[f"swap{index}"] = net(index, x, bev, feat, I_inv, E_inv)
    att = Attention(128, dim_head=32, dropout=0.1, window_size=8).eval()
    rs = np.random.RandomState(165)
    asd = {"to_qkv.weight": torch.from_numpy(rs.uniform(-0.09, 0.09, (384, 128)).astype(np.float32)),
           "to_out.0.weight": torch.from_numpy(rs.uniform(-0.09, 0.09, (128, 128)).astype(np.float32)),
           "rel_pos_bias.weight": torch.from_numpy(rs.standard_normal((225, 4)).astype(np.float32))}
    att.load_state_dict(asd, strict=True)
    xa = torch.from_numpy(rs.standard_normal((2, 128, 8, 8)).astype(np.float32))
    out["self_attn"] = att(xa)
    assert torch.equal(att.rel_pos_indices, FO.rel_pos_indices(8))
    down = torch.nn.Sequential(torch.nn.Sequential(
        torch.nn.Conv2d(128, 32, 3, 1, 1, bias=False), torch.nn.PixelUnshuffle(2), torch.nn.Conv2d(128, 128, 3, padding=1, bias=False),
        torch.nn.BatchNorm2d(128), torch.nn.ReLU(inplace=True), torch.nn.Conv2d(128, 128, 1, padding=0, bias=False),
        torch.nn.BatchNorm2d(128))).eval()                       # fax_modules.py:478-492 for dim[i] = dim[i+1] = 128
    dsd = {}
    for k, v in down.state_dict().items():
        if "num_batches" in k:
            dsd[k] = v
        elif "running_var" in k:
            dsd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, tuple(v.shape)).astype(np.float32))
        else:
            dsd[k] = torch.from_numpy((0.1 * rs.standard_normal(tuple(v.shape))).astype(np.float32)) + (1.0 if k.endswith((".3.weight", ".6.weight")) else 0.0)
    down.load_state_dict(dsd, strict=True)
    xd = torch.from_numpy(rs.standard_normal((2, 128, 8, 8)).astype(np.float32))
    out["down"] = down(xd)
    save("g16_fax.npz", **out, **{f"down_sd.{k}": v for k, v in dsd.items() if "num_batches" not in k}, down_x=xd, attn_x=xa,
         **{f"attn_sd.{k}": v for k, v in asd.items()})


def g15_compressor():
    from opencood.models.sub_modules.naive_compress import NaiveCompressor
    net = NaiveCompressor(256, 4).eval()
    net.load_state_dict(DO.compressor_state_dict(256, 4, seed=151), strict=True)
    x = torch.from_numpy(np.random.RandomState(152).standard_normal((3, 256, 10, 12)).astype(np.float32))
    save("g15_compressor.npz", seed_weights=151, seed_x=152, out=net(x))


def g14_loss():
    from opencood.loss.point_pillar_loss import PointPillarLoss
    torch.set_grad_enabled(True)
    psm, rm, tgt = loss_inputs()
    psm.requires_grad_(True)
    rm.requires_grad_(True)
    crit = PointPillarLoss({"cls_weight": 1.0, "reg": 2.0})
    total = crit({"psm": psm, "rm": rm}, tgt)
    total.backward()
    save("g14_loss.npz", total=total.detach(), reg=crit.loss_dict["reg_loss"].detach(), conf=crit.loss_dict["conf_loss"].detach(),
         d_psm=psm.grad, d_rm=rm.grad)
    torch.set_grad_enabled(False)


def g7_pointpillar():
    """PointPillar.forward (return_features), eval-mode BN with non-trivial running stats:
    2 agents x 400 pillars on a 64 x 48 canvas -> (2, 256, 12, 16); also the PFN output."""
    from opencood.models.point_pillar import PointPillar
    args = PO.make_args(64, 48)
    net = PointPillar(args).eval()
    net.set_return_features()
    net.load_state_dict(PO.random_state_dict(args, seed=71), strict=True)
    vf, vc, vn = PO.synthetic_pillars(2, 400, 64, 48, args, seed=72)
    batch = {"voxel_features": vf, "voxel_coords": vc, "voxel_num_points": vn}
    pf = net.pillar_vfe(dict(batch))["pillar_features"]
    y = net({"processed_lidar": batch})
    save("g7_pointpillar.npz", seed_weights=71, seed_pillars=72, grid=np.array([64, 48]), n_agents=np.array(2),
         n_per_agent=np.array(400), pillar_features=pf, out=y)


def g9_model():
    """BevformerPointPillarHetero.forward on a LiDAR-only batch (the camera branch is never entered;
    BEVFormerWrapper is replaced by an empty module so that the reference class can be constructed)."""
    import opencood.models.bevformer_point_pillar_hetero as M

    class NoCamera(torch.nn.Module):
        def __init__(self, cfg):
            super().__init__()

        def set_return_features(self):
            pass

    M.BEVFormerWrapper = NoCamera
    cfg = model_config()
    net = M.BevformerPointPillarHetero(cfg).eval()
    sd = model_state_dict(cfg, 91)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out = net(model_batch(cfg, 95))
    save("g9_model.npz", seed_weights=91, seed_batch=95, psm=out["psm"], rm=out["rm"])


def g8_decoder():
    """HeteroDecoder.forward (use_upsample=False), eval-mode BN, B=3 with ego types 1, 0, 1."""
    from opencood.models.sub_modules.hetero_decoder import HeteroDecoder
    params = DO.make_params()
    net = HeteroDecoder(params).eval()
    net.load_state_dict(DO.random_state_dict(params, seed=81), strict=True)
    rs = np.random.RandomState(82)
    x = torch.from_numpy(rs.standard_normal((3, 1, 256, 12, 10)).astype(np.float32))
    mode = torch.tensor([[1, 0], [0, 1], [1, 1]])
    psm, rm = net(x, mode, use_upsample=False)
    save("g8_decoder.npz", seed_weights=81, seed_x=82, mode=mode, psm=psm, rm=rm)


def g10_postprocess():
    """Reference post-processing + AP with shapely.Polygon replaced by the oracle's convex clipper."""
    import re
    from oracle import postprocess_oracle as PPO
    gt_line = [l for l in open("/root/reference/opencood/data_utils/datasets/__init__.py") if l.startswith("GT_RANGE")][0]
    gt_range = json.loads(re.search(r"\[.*?\]", gt_line).group(0))
    assert gt_range == PPO.GT_RANGE, gt_range
    _stub("opencood.visualization"); _stub("opencood.visualization.vis_utils")
    _stub("opencood.utils.box_overlaps", bbox_overlaps=None)
    _stub("cv2"); _stub("mmcv", Config=object, DictAction=object)
    _stub("opencood.data_utils.datasets", GT_RANGE=gt_range)   # the real package pulls open3d / cv2 in
    import opencood.utils.common_utils as cu
    cu.Polygon = PPO.Polygon
    from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
    from opencood.utils import eval_utils

    params = PPO.make_params(W=96, H=64)
    pp = VoxelPostprocessor(params, train=False)
    anchors = pp.generate_anchor_box()
    assert np.array_equal(anchors, PPO.generate_anchor_box(params))
    T1 = np.eye(4, dtype=np.float32)
    c, s = np.cos(0.3), np.sin(0.3)
    T1[:2, :2] = [[c, -s], [s, c]]
    T1[:3, 3] = [5.0, -3.0, 0.2]
    stat = {t: {"tp": [], "fp": [], "gt": 0} for t in (0.3, 0.5, 0.7)}
    frames = []
    for seed0 in (101, 103):
        psm0, rm0, _, gt = PPO.synthetic_heads(params, seed=seed0, n_obj=14)
        psm1, rm1, _, _ = PPO.synthetic_heads(params, seed=seed0 + 1, n_obj=6)
        data = {"ego": {"anchor_box": torch.from_numpy(anchors), "transformation_matrix": torch.eye(4)},
                "7": {"anchor_box": torch.from_numpy(anchors), "transformation_matrix": torch.from_numpy(T1)}}
        out = {"ego": {"psm": torch.from_numpy(psm0), "rm": torch.from_numpy(rm0)},
               "7": {"psm": torch.from_numpy(psm1), "rm": torch.from_numpy(rm1)}}
        boxes, scores = pp.post_process(data, out)
        frames.append((boxes.numpy(), scores.numpy()))
        for t in stat:
            eval_utils.caluclate_tp_fp(boxes, scores, torch.from_numpy(gt), stat, t)
    tp = {t: list(stat[t]["tp"]) for t in stat}
    fp = {t: list(stat[t]["fp"]) for t in stat}
    ap = {t: eval_utils.calculate_ap(stat, t)[0] for t in stat}
    save("g10_postprocess.npz", seeds=np.array([101, 103]), T1=T1,
         boxes0=frames[0][0], scores0=frames[0][1], boxes1=frames[1][0], scores1=frames[1][1],
         tp30=np.array(tp[0.3]), tp50=np.array(tp[0.5]), tp70=np.array(tp[0.7]),
         fp30=np.array(fp[0.3]), fp50=np.array(fp[0.5]), fp70=np.array(fp[0.7]),
         ap=np.array([ap[0.3], ap[0.5], ap[0.7]]), gt_total=np.array(stat[0.5]["gt"]))
    print("g10: boxes", frames[0][0].shape, frames[1][0].shape, "AP", ap)


def g17_labels():
    """Reference VoxelPostprocessor.generate_label; its Cython bbox_overlaps (utils/box_overlaps.pyx, not compiled here) is
    replaced by a numpy loop with the same arithmetic (float32, + 1 on widths / heights, zero unless both overlaps > 0)."""
    import re
    from make_goldens_inputs import label_inputs, label_params

    def bbox_overlaps(boxes, query):
        out = np.zeros((len(boxes), len(query)), np.float32)
        for k in range(len(query)):
            qa = np.float32((query[k, 2] - query[k, 0] + 1) * (query[k, 3] - query[k, 1] + 1))
            iw = np.minimum(boxes[:, 2], query[k, 2]) - np.maximum(boxes[:, 0], query[k, 0]) + np.float32(1)
            ih = np.minimum(boxes[:, 3], query[k, 3]) - np.maximum(boxes[:, 1], query[k, 1]) + np.float32(1)
            ua = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1) + qa - iw * ih
            ok = (iw > 0) & (ih > 0)
            out[ok, k] = (iw * ih / ua)[ok]
        return out

    gt_line = [l for l in open("/root/reference/opencood/data_utils/datasets/__init__.py") if l.startswith("GT_RANGE")][0]
    _stub("opencood.visualization"); _stub("opencood.visualization.vis_utils")
    _stub("opencood.utils.box_overlaps", bbox_overlaps=bbox_overlaps)
    _stub("cv2"); _stub("mmcv", Config=object, DictAction=object)
    _stub("opencood.data_utils.datasets", GT_RANGE=json.loads(re.search(r"\[.*?\]", gt_line).group(0)))
    from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
    params = label_params()
    pp = VoxelPostprocessor(params, train=True)
    anchors = pp.generate_anchor_box()
    out = {}
    for tag, seed in (("a", 171), ("b", 172)):
        gt, mask = label_inputs(seed)
        lab = pp.generate_label(gt_box_center=gt, anchors=anchors, mask=mask)
        out.update({f"pos_{tag}": lab["pos_equal_one"].astype(np.uint8), f"neg_{tag}": lab["neg_equal_one"].astype(np.uint8),
                    f"targets_{tag}": lab["targets"]})
        print("g17", tag, "positives", int(lab["pos_equal_one"].sum()), "negatives", int(lab["neg_equal_one"].sum()))
    # the collate of two samples
    col = pp.collate_batch([{"pos_equal_one": out["pos_a"], "neg_equal_one": out["neg_a"], "targets": out["targets_a"]},
                            {"pos_equal_one": out["pos_b"], "neg_equal_one": out["neg_b"], "targets": out["targets_b"]}])
    assert tuple(col["targets"].shape) == (2,) + out["targets_a"].shape
    save("g17_labels.npz", seeds=np.array([171, 172]), **out)


def g11_cross_view():
    """CrossViewAttention + BEVEmbedding.grid of the reference, seeded weights / inputs from oracle/cvt_oracle.py."""
    from oracle import cvt_oracle as CO
    _stub("torchvision"); _stub("torchvision.models"); _stub("torchvision.models.resnet", Bottleneck=object)
    from opencood.models.sub_modules.cvt_modules import BEVEmbedding, CrossViewAttention
    out = {}
    for tag, no_feat, skip in (("a", False, True), ("b", True, False)):
        cfg = CO.make_config()
        cfg["no_image_features"], cfg["skip"] = no_feat, skip
        net = CrossViewAttention(12, 12, 64, 128, cfg).eval()
        sd = CO.random_state_dict(64, 128, cfg, seed=111)
        own = net.state_dict()
        load = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in load and "num_batches_tracked" not in k]
        assert not missing, missing
        net.load_state_dict(load, strict=False)
        bev = BEVEmbedding(128, 1.0, 64, 64, 100.0, 100.0, 0.0, [128, 128, 64])
        assert torch.allclose(bev.grid, CO.bev_grid(64, 64, 100.0, 100.0, 0.0, 3))
        x, feat, I_inv, E_inv = CO.synthetic_inputs(2, 4, 64, 12, 12, 128, 8, 8, seed=112)
        with torch.no_grad():
            y = net(x, bev, feat, I_inv, E_inv)
        out["y_" + tag] = y
    save("g11_cross_view.npz", seed_weights=111, seed_inputs=112, **out)


if __name__ == "__main__":
    if "g11" in sys.argv[1:]:
        g11_cross_view()
        sys.exit(0)
    if "g17" in sys.argv[1:]:
        g17_labels()
        sys.exit(0)
    if "g16" in sys.argv[1:]:
        g16_fax()
        sys.exit(0)
    if "g15" in sys.argv[1:]:
        g15_compressor()
        sys.exit(0)
    if "g14" in sys.argv[1:]:
        g14_loss()
        sys.exit(0)
    if "g12" in sys.argv[1:]:
        g12_fusion_cfg2()
        sys.exit(0)
    if "g13" in sys.argv[1:]:
        g13_fusion_cfg3()
        sys.exit(0)
    if "g18" in sys.argv[1:]:
        g18_fusion_cfg4()
        sys.exit(0)
    if "g10" in sys.argv[1:]:
        g10_postprocess()
        sys.exit(0)
    g9_model()
    g8_decoder()
    g7_pointpillar()
    g1_attention()
    g2_warp()
    g3_block("sequential")
    g3_block("parallel")
    g4_fusion_c256()
    g5_fusion_ragged()
    g6_fusion_cfg1()
