"""CPU restatement (plain PyTorch fp32) of the reference's FAX camera -> BEV lift (opencood/models/sub_modules/fax_modules.py):
CrossWinAttention (:183-252), CrossViewSwapAttention (:255-445), Attention (:96-180), the down-sampling blocks and the
level loop of FAXModule (:448-525).  TEST INFRASTRUCTURE ONLY.

Parity: CrossViewSwapAttention, Attention and the down-sampling block are PINNED by tests/golden/g16_fax.npz (the reference
modules imported with torchvision stubbed).  The ResNetBottleNeck layers of FAXModule are torchvision's Bottleneck (absent,
version unpinned): restated from its published definition in oracle/camera_oracle.py, so the assembled module is parity
UNPINNED at those layers, exactly like the CVT branch.  Eval mode: BatchNorm uses running statistics, Dropout is the identity;
under ``cvt_oracle.batch_statistics()`` the BatchNorms use batch statistics and update their running buffers (the training-mode
restatement the gradient tests differentiate, in float64).  Every function follows the dtype of its inputs."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from . import cvt_oracle as CO
from .cvt_oracle import generate_grid


def _bn(x, sd, p):
    tr = CO.BN_TRAINING[0]
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"], tr, 0.1 if tr else 0.0, 1e-5)


def bev_grids(bev_height, bev_width, h_meters, w_meters, offset, upsample_scales) -> List[Tensor]:
    """BEVEmbedding.grid{i} (fax_modules.py:66-83): (3, h, w) ego-frame coordinates of the BEV cells of every level."""
    sh, sw = bev_height / h_meters, bev_width / w_meters
    V = torch.tensor([[0.0, -sw, bev_width / 2.0], [-sh, 0.0, bev_height * offset + bev_height / 2.0], [0.0, 0.0, 1.0]])
    out = []
    for scale in upsample_scales:
        h, w = bev_height // scale, bev_width // scale
        grid = generate_grid(h, w).squeeze(0)
        grid[0] = bev_width * grid[0]
        grid[1] = bev_height * grid[1]
        out.append((V.inverse() @ grid.reshape(3, -1)).reshape(3, h, w))
    return out


def _ln_linear(x, sd, p, dim):
    y = F.layer_norm(x, (dim,), sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], 1e-5)
    return F.linear(y, sd[f"{p}.1.weight"], sd.get(f"{p}.1.bias"))


def cross_win_attention(q, k, v, skip, sd: Dict[str, Tensor], p: str, heads: int, dim_head: int):
    """CrossWinAttention.forward (fax_modules.py:205-252).  q (b n X Y W1 W2 d), k / v (b n x y w1 w2 d), skip (b X Y W1 W2 d)
    -> (b X Y W1 W2 d): inside window l every query of every camera attends to the keys of all cameras in window l; the
    per-camera results are averaged."""
    b, n, X, Y, W1, W2, dim = q.shape               # n query "cameras" (1 when the level has no BEV embedding)
    _, nk, x, y, w1, w2, _ = k.shape
    assert X * Y == x * y
    q = q.permute(0, 2, 3, 1, 4, 5, 6).reshape(b, X * Y, n * W1 * W2, dim)
    k = k.permute(0, 2, 3, 1, 4, 5, 6).reshape(b, x * y, nk * w1 * w2, dim)
    v = v.permute(0, 2, 3, 1, 4, 5, 6).reshape(b, x * y, nk * w1 * w2, dim)
    q, k, v = _ln_linear(q, sd, f"{p}.to_q", dim), _ln_linear(k, sd, f"{p}.to_k", dim), _ln_linear(v, sd, f"{p}.to_v", dim)
    split = lambda t: t.reshape(b, t.shape[1], t.shape[2], heads, dim_head).permute(0, 3, 1, 2, 4)    # b m l Q d
    q, k, v = split(q), split(k), split(v)
    dot = (dim_head ** -0.5) * torch.einsum("bmlqd,bmlkd->bmlqk", q, k)
    a = torch.einsum("bmlqk,bmlkd->bmlqd", dot.softmax(dim=-1), v)
    a = a.permute(0, 2, 3, 1, 4).reshape(b, X, Y, n, W1, W2, heads * dim_head).permute(0, 3, 1, 2, 4, 5, 6)
    z = F.linear(a, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"]).mean(1)
    return z if skip is None else z + skip


def _bn_relu_conv(x, sd, p):
    return F.conv2d(F.relu(_bn(x, sd, f"{p}.0")), sd[f"{p}.2.weight"])


def _pad_divisible(x, win_h, win_w):
    h, w = x.shape[-2:]
    padh = ((h + win_h) // win_h) * win_h - h if h % win_h else 0
    padw = ((w + win_w) // win_w) * win_w - w if w % win_w else 0
    return F.pad(x, (0, padw, 0, padh), value=0)


def _win(t, w1, w2):      # b n d (x w1) (y w2) -> b n x y w1 w2 d
    b, n, d, H, W = t.shape
    return t.reshape(b, n, d, H // w1, w1, W // w2, w2).permute(0, 1, 3, 5, 4, 6, 2)


def _grid(t, w1, w2):     # b n d (w1 x) (w2 y) -> b n x y w1 w2 d
    b, n, d, H, W = t.shape
    return t.reshape(b, n, d, w1, H // w1, w2, W // w2).permute(0, 1, 4, 6, 3, 5, 2)


def cross_view_swap_attention(x, grid, feature, I_inv, E_inv, sd: Dict[str, Tensor], cfg: dict, index: int):
    """CrossViewSwapAttention.forward (fax_modules.py:325-445).  x (b, dim, H, W); grid (3, H, W) = BEVEmbedding.grid{index};
    feature (b, n, feat_dim, h, w); I_inv (b, n, 3, 3); E_inv (b, n, 4, 4).  cfg: cross_view + cross_view_swap keys."""
    b, n, feat_dim, h, w = feature.shape
    _, dim, H, W = x.shape
    heads, dim_head = cfg["heads"][index], cfg["dim_head"][index]
    qw, fw = cfg["q_win_size"][index], cfg["feat_win_size"][index]
    pixel = generate_grid(h, w)[None].to(x.dtype)
    grid = grid.to(x.dtype)
    pixel[:, :, 0] *= cfg["image_width"]
    pixel[:, :, 1] *= cfg["image_height"]
    c = E_inv[..., -1:].reshape(b * n, 4, 1, 1)
    c_embed = F.conv2d(c, sd["cam_embed.weight"])
    cam = I_inv @ pixel.reshape(1, 1, 3, h * w)
    cam = F.pad(cam, (0, 0, 0, 1), value=1)
    d = (E_inv @ cam).reshape(b * n, 4, h, w)
    img_embed = F.conv2d(d, sd["img_embed.weight"]) - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    if cfg["bev_embedding_flag"][index]:
        w_embed = F.conv2d(grid[:2][None], sd["bev_embed.weight"], sd["bev_embed.bias"])
        bev_embed = w_embed - c_embed
        bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
        query = bev_embed.reshape(b, n, dim, H, W) + x[:, None]
    else:
        query = x[:, None].expand(b, 1, dim, H, W)          # a single "camera" of queries (x[:, None], fax_modules.py:393)
    feature_flat = feature.reshape(b * n, feat_dim, h, w)
    key_flat = img_embed + _bn_relu_conv(feature_flat, sd, "feature_proj") if not cfg["no_image_features"] else img_embed
    val_flat = _bn_relu_conv(feature_flat, sd, "feature_linear")
    key = _pad_divisible(key_flat.reshape(b, n, dim, h, w), fw[0], fw[1])
    val = _pad_divisible(val_flat.reshape(b, n, dim, h, w), fw[0], fw[1])
    skip = x.reshape(b, dim, H // qw[0], qw[0], W // qw[1], qw[1]).permute(0, 2, 4, 3, 5, 1) if cfg["skip"] else None
    # local-to-local
    q1 = cross_win_attention(_win(query, qw[0], qw[1]), _win(key, fw[0], fw[1]), _win(val, fw[0], fw[1]), skip, sd,
                             "cross_win_attend_1", heads, dim_head)
    q1 = q1.permute(0, 1, 3, 2, 4, 5).reshape(b, H, W, dim)

    def mlp(t, i):
        z = F.layer_norm(t, (dim,), sd[f"prenorm_{i}.weight"], sd[f"prenorm_{i}.bias"], 1e-5)
        return t + F.linear(F.gelu(F.linear(z, sd[f"mlp_{i}.0.weight"], sd[f"mlp_{i}.0.bias"])), sd[f"mlp_{i}.2.weight"], sd[f"mlp_{i}.2.bias"])

    q1 = mlp(q1, 1)
    x_skip = q1
    # local-to-global: queries stay in windows, keys / values are re-partitioned as a dilated grid
    qn = q1.permute(0, 3, 1, 2)[:, None].expand(b, n, dim, H, W)
    skip2 = x_skip.reshape(b, H // qw[0], qw[0], W // qw[1], qw[1], dim).permute(0, 1, 3, 2, 4, 5) if cfg["skip"] else None
    q2 = cross_win_attention(_win(qn, qw[0], qw[1]), _grid(key, fw[0], fw[1]), _grid(val, fw[0], fw[1]), skip2, sd,
                             "cross_win_attend_2", heads, dim_head)
    q2 = q2.permute(0, 1, 3, 2, 4, 5).reshape(b, H, W, dim)
    q2 = mlp(q2, 2)
    q2 = F.layer_norm(q2, (dim,), sd["postnorm.weight"], sd["postnorm.bias"], 1e-5)
    return q2.permute(0, 3, 1, 2)


def rel_pos_indices(window_size: int) -> Tensor:
    pos = torch.arange(window_size)
    grid = torch.stack(torch.meshgrid(pos, pos, indexing="ij")).reshape(2, -1).t()
    rel = grid[:, None] - grid[None, :] + window_size - 1
    return (rel * torch.tensor([2 * window_size - 1, 1])).sum(-1)


def self_attention(x, sd: Dict[str, Tensor], dim_head: int, window_size: int, p: str = ""):
    """Attention.forward (fax_modules.py:136-180): full self-attention over the (h, w) map with a relative-position bias."""
    pre = f"{p}." if p else ""
    b, dim, h, w = x.shape
    m = dim // dim_head
    t = x.permute(0, 2, 3, 1).reshape(b, h * w, dim)
    q, k, v = F.linear(t, sd[f"{pre}to_qkv.weight"]).chunk(3, dim=-1)
    split = lambda u: u.reshape(b, h * w, m, dim_head).permute(0, 2, 1, 3)
    q, k, v = split(q) * dim_head ** -0.5, split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k)
    bias = sd[f"{pre}rel_pos_bias.weight"][rel_pos_indices(window_size)]
    sim = sim + bias.permute(2, 0, 1)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, h, w, dim)
    return F.linear(out, sd[f"{pre}to_out.0.weight"]).permute(0, 3, 1, 2)


def downsample_block(x, sd: Dict[str, Tensor], p: str):
    """FAXModule.downsample_layers[i] (fax_modules.py:478-492): conv3x3 (no bias) -> PixelUnshuffle(2) -> conv3x3 -> BN -> ReLU
    -> conv1x1 -> BN."""
    q = f"{p}.0"
    y = F.conv2d(x, sd[f"{q}.0.weight"], None, 1, 1)
    y = F.pixel_unshuffle(y, 2)
    y = F.conv2d(y, sd[f"{q}.2.weight"], None, 1, 1)
    y = F.relu(_bn(y, sd, f"{q}.3"))
    y = F.conv2d(y, sd[f"{q}.5.weight"])
    return _bn(y, sd, f"{q}.6")


def fax_module(features: List[Tensor], intrinsic, extrinsic, sd: Dict[str, Tensor], cfg: dict):
    """FAXModule.forward (fax_modules.py:499-525).  features: per level (N, n, C_i, h_i, w_i) for N = b l agents;
    intrinsic (N, n, 3, 3), extrinsic (N, n, 4, 4) -> (N, dim[-1], H, W).  The ResNetBottleNeck layers follow
    oracle/camera_oracle.bottleneck (torchvision arithmetic, unpinned)."""
    from .camera_oracle import bottleneck
    N = features[0].shape[0]
    I_inv = intrinsic.inverse()
    E_inv = extrinsic
    be = cfg["bev_embedding"]
    grids = bev_grids(be["bev_height"], be["bev_width"], be["h_meters"], be["w_meters"], be["offset"], be["upsample_scales"])
    cv = dict(cfg["cross_view"])
    cv.update(cfg["cross_view_swap"])
    x = sd["bev_embedding.learned_features"][None].repeat(N, 1, 1, 1)
    for i, feature in enumerate(features):
        sub = {k[len(f"cross_views.{i}."):]: v for k, v in sd.items() if k.startswith(f"cross_views.{i}.")}
        x = cross_view_swap_attention(x, grids[i], feature, I_inv, E_inv, sub, cv, i)
        for j in range(cfg["middle"][i]):
            x = bottleneck(x, sd, f"layers.{i}.{j}")
        if i < len(features) - 1:
            x = downsample_block(x, sd, f"downsample_layers.{i}")
    sa = cfg["self_attn"]
    return self_attention(x, sd, sa["dim_head"], sa["window_size"], "self_attn")


# ---- seeded configs / weights / inputs (numpy legacy stream, shared by goldens and tests) ----
def make_swap_config(image=64):
    return {"image_height": image, "image_width": image, "no_image_features": False, "skip": True, "heads": [4, 4], "dim_head": [32, 32],
            "qkv_bias": True, "rel_pos_emb": False, "q_win_size": [[8, 8], [4, 4]], "feat_win_size": [[4, 4], [2, 2]],
            "bev_embedding_flag": [True, False]}


def _rs(seed):
    return np.random.RandomState(seed)


def _t(a):
    return torch.from_numpy(np.asarray(a, np.float32))


def swap_state_dict(feat_dim: int, dim: int, cfg: dict, index: int, seed: int = 0) -> Dict[str, Tensor]:
    """Reference-named weights of one CrossViewSwapAttention (fax_modules.py:276-323)."""
    rs = _rs(seed)
    sd: Dict[str, Tensor] = {}

    def bn(name, c):
        sd[f"{name}.weight"] = _t(1 + 0.2 * rs.standard_normal(c)); sd[f"{name}.bias"] = _t(0.2 * rs.standard_normal(c))
        sd[f"{name}.running_mean"] = _t(0.3 * rs.standard_normal(c)); sd[f"{name}.running_var"] = _t(rs.uniform(0.5, 1.5, c))

    def lin(name, co, ci, bias=True, shape=None):
        b = 1.0 / np.sqrt(ci)
        sd[f"{name}.weight"] = _t(rs.uniform(-b, b, shape or (co, ci)))
        if bias:
            sd[f"{name}.bias"] = _t(rs.uniform(-b, b, co))

    def ln(name):
        sd[f"{name}.weight"] = _t(1 + 0.1 * rs.standard_normal(dim)); sd[f"{name}.bias"] = _t(0.1 * rs.standard_normal(dim))

    for p in ("feature_linear", "feature_proj"):
        bn(f"{p}.0", feat_dim)
        lin(f"{p}.2", dim, feat_dim, bias=False, shape=(dim, feat_dim, 1, 1))
    if cfg["bev_embedding_flag"][index]:
        lin("bev_embed", dim, 2, shape=(dim, 2, 1, 1))
    lin("img_embed", dim, 4, bias=False, shape=(dim, 4, 1, 1))
    lin("cam_embed", dim, 4, bias=False, shape=(dim, 4, 1, 1))
    hd = cfg["heads"][index] * cfg["dim_head"][index]
    for a in ("cross_win_attend_1", "cross_win_attend_2"):
        for t in ("to_q", "to_k", "to_v"):
            ln(f"{a}.{t}.0")
            lin(f"{a}.{t}.1", hd, dim, bias=cfg["qkv_bias"])
        lin(f"{a}.proj", dim, hd)
    for i in (1, 2):
        ln(f"prenorm_{i}")
        lin(f"mlp_{i}.0", 2 * dim, dim)
        lin(f"mlp_{i}.2", dim, 2 * dim)
    ln("postnorm")
    return sd


def synthetic_inputs(b, n, feat_dim, h, w, dim, H, W, seed=0, image=64):
    rs = _rs(seed)
    x = _t(rs.standard_normal((b, dim, H, W)))
    feature = _t(rs.standard_normal((b, n, feat_dim, h, w)))
    f = 0.8 * image
    I = np.tile(np.array([[f, 0, image / 2], [0, f, image / 2], [0, 0, 1]], np.float32), (b, n, 1, 1))
    I_inv = torch.from_numpy(np.linalg.inv(I)).float()
    E = np.zeros((b, n, 4, 4), np.float32)
    for bi in range(b):
        for ni in range(n):
            yaw = 2 * np.pi * ni / n + 0.1 * rs.standard_normal()
            c, s = np.cos(yaw), np.sin(yaw)
            E[bi, ni] = np.array([[c, -s, 0, 1.5 * c], [s, c, 0, 1.5 * s], [0, 0, 1, 1.6], [0, 0, 0, 1]], np.float32)
    return x, feature, I_inv, torch.from_numpy(E)


def fax_camera_encoder(batch: dict, sd, cfg: dict) -> Tensor:
    """FaxFusedTransformer's camera branch (fax_fused_transformer.py:37-57): images (N, M, H, W, 3), intrinsic (N, M, 3, 3),
    extrinsic (N, M, 4, 4) -> BEV features (N, C, Hb, Wb).  ResNet and Bottleneck arithmetic: oracle/camera_oracle.py (unpinned)."""
    from . import camera_oracle as CAM
    cam = batch["camera"][:, None]                                 # camera.unsqueeze(1): b = N agents, l = 1
    feats = CAM.resnet_encoder(cam, sd, cfg["encoder"], prefix="encoder.encoder")
    fsd = {k[len("fax."):]: v for k, v in sd.items() if k.startswith("fax.")}
    x = fax_module([f[:, 0] for f in feats], batch["intrinsic"], batch["extrinsic"], fsd, cfg["fax"])
    return CAM.naive_decoder_up(x, sd, "decoder", cfg["decoder"]["num_layer"])


def make_camera_config(image=64, num_layers=18):
    """A reduced FAX camera encoder with the structure of opcl/fax_point_pillar_v2xt.yaml (camera block): ResNet pyramid levels
    1..3, dim 128 at every level, BEV queries bev/2, bev/4, bev/8 with bev = 64, q windows [8, 8, 8], feature windows [2, 2, 2]..."""
    enc = {"num_layers": num_layers, "pretrained": False, "image_height": image, "image_width": image, "id_pick": [1, 2, 3]}
    fax = {"dim": [128, 128, 128], "middle": [1, 1, 1],
           "bev_embedding": {"sigma": 1.0, "bev_height": 64, "bev_width": 64, "h_meters": 100.0, "w_meters": 100.0, "offset": 0.0,
                             "upsample_scales": [2, 4, 8]},
           "cross_view": {"image_height": image, "image_width": image, "no_image_features": False, "skip": True, "heads": [4, 4, 4],
                          "dim_head": [32, 32, 32], "qkv_bias": True},
           "cross_view_swap": {"rel_pos_emb": False, "q_win_size": [[8, 8], [8, 8], [8, 8]],
                               "feat_win_size": [[2, 2], [2, 2], [2, 2]], "bev_embedding_flag": [True, False, False]},
           "self_attn": {"dim_head": 32, "dropout": 0.1, "window_size": 8}}
    return {"encoder": enc, "fax": fax, "decoder": {"input_dim": 128, "num_layer": 2, "num_ch_dec": [256, 256]}, "anchor_number": 2}
