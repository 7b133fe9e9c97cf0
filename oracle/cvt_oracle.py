"""CPU restatement (plain PyTorch fp32) of the camera -> BEV lift of the reference's CVT encoder: CrossViewAttention and
CrossAttention (opencood/models/sub_modules/cvt_modules.py:95-280), with the grids of generate_grid / BEVEmbedding
(:15-91).  TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/g11_cross_view.npz (the reference module imported with
torchvision stubbed: its ResNet bottlenecks are not part of this function).  Eval mode: BatchNorm2d uses running statistics."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


def generate_grid(height: int, width: int) -> Tensor:
    """cvt_modules.py:15-26, including its argument order: meshgrid((xs, ys)) with 'ij' indexing gives (width, height) maps."""
    xs = torch.linspace(0, 1, width)
    ys = torch.linspace(0, 1, height)
    yy, xx = torch.meshgrid((xs, ys), indexing="ij")
    indices = torch.stack([xx, yy], 0)
    indices = F.pad(indices, (0, 0, 0, 0, 0, 1), value=1)
    return indices[None]


def bev_grid(bev_height, bev_width, h_meters, w_meters, offset, n_decoder_blocks) -> Tensor:
    """BEVEmbedding.grid (cvt_modules.py:66-86): (3, h, w) ego-frame coordinates of the BEV query cells."""
    h = bev_height // (2 ** n_decoder_blocks)
    w = bev_width // (2 ** n_decoder_blocks)
    grid = generate_grid(h, w).squeeze(0)
    grid[0] = bev_width * grid[0]
    grid[1] = bev_height * grid[1]
    sh, sw = bev_height / h_meters, bev_width / w_meters
    V = torch.tensor([[0.0, -sw, bev_width / 2.0], [-sh, 0.0, bev_height * offset + bev_height / 2.0], [0.0, 0.0, 1.0]])
    g = V.inverse() @ grid.reshape(3, -1)
    return g.reshape(3, h, w)


# training-mode restatement (tests of the camera branch's backward pass): BatchNorm on batch statistics, running buffers of the
# state dict updated in place with nn.BatchNorm2d's default momentum 0.1 - switched on by the tests through batch_statistics()
BN_TRAINING = [False]


class batch_statistics:
    def __enter__(self):
        BN_TRAINING[0] = True

    def __exit__(self, *a):
        BN_TRAINING[0] = False


def _bn_relu_conv(x, sd, p):
    y = F.batch_norm(x, sd[f"{p}.0.running_mean"], sd[f"{p}.0.running_var"], sd[f"{p}.0.weight"], sd[f"{p}.0.bias"],
                     BN_TRAINING[0], 0.1 if BN_TRAINING[0] else 0.0, 1e-5)
    return F.conv2d(F.relu(y), sd[f"{p}.2.weight"])


def _ln_linear(x, sd, p, dim):
    y = F.layer_norm(x, (dim,), sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], 1e-5)
    return F.linear(y, sd[f"{p}.1.weight"], sd.get(f"{p}.1.bias"))


def cross_attention(q, k, v, skip, sd: Dict[str, Tensor], heads: int, dim_head: int, p="cross_attend"):
    """CrossAttention.forward (cvt_modules.py:118-173).  q (b n d H W), k / v (b n d h w)."""
    b, n, dim, H, W = q.shape
    q = q.permute(0, 1, 3, 4, 2).reshape(b, n, H * W, dim)
    k = k.permute(0, 1, 3, 4, 2).reshape(b, n, -1, dim)
    v = v.permute(0, 1, 3, 4, 2).reshape(b, -1, dim)
    q = _ln_linear(q, sd, f"{p}.to_q", dim)
    k = _ln_linear(k, sd, f"{p}.to_k", dim)
    v = _ln_linear(v, sd, f"{p}.to_v", dim)
    m, d = heads, dim_head
    q = q.reshape(b, n, H * W, m, d).permute(0, 3, 1, 2, 4)           # b m n Q d
    k = k.reshape(b, n, -1, m, d).permute(0, 3, 1, 2, 4)              # b m n K d
    v = v.reshape(b, -1, m, d).permute(0, 2, 1, 3)                    # b m (n K) d
    dot = (d ** -0.5) * torch.einsum("bmnqd,bmnkd->bmnqk", q, k)
    dot = dot.permute(0, 1, 3, 2, 4).reshape(b, m, H * W, -1)         # b m Q (n K)
    att = dot.softmax(dim=-1)
    a = torch.einsum("bmqk,bmkd->bmqd", att, v)
    a = a.permute(0, 2, 1, 3).reshape(b, H * W, m * d)
    z = F.linear(a, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])
    if skip is not None:
        z = z + skip.permute(0, 2, 3, 1).reshape(b, H * W, dim)
    z = F.layer_norm(z, (dim,), sd[f"{p}.prenorm.weight"], sd[f"{p}.prenorm.bias"], 1e-5)
    hdn = F.gelu(F.linear(z, sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"]))
    z = z + F.linear(hdn, sd[f"{p}.mlp.2.weight"], sd[f"{p}.mlp.2.bias"])
    z = F.layer_norm(z, (dim,), sd[f"{p}.postnorm.weight"], sd[f"{p}.postnorm.bias"], 1e-5)
    return z.reshape(b, H, W, dim).permute(0, 3, 1, 2)


def cross_view_attention(x, grid, feature, I_inv, E_inv, sd: Dict[str, Tensor], cfg: dict):
    """CrossViewAttention.forward (cvt_modules.py:216-280).  x (b, dim, H, W); grid (3, H, W) = BEVEmbedding.grid;
    feature (b, n, feat_dim, h, w); I_inv (b, n, 3, 3); E_inv (b, n, 4, 4)."""
    b, n, feat_dim, h, w = feature.shape
    pixel = generate_grid(h, w)[None].to(x.dtype)
    pixel[:, :, 0] *= cfg["image_width"]
    pixel[:, :, 1] *= cfg["image_height"]
    c = E_inv[..., -1:]
    c_flat = c.reshape(b * n, 4, 1, 1)
    c_embed = F.conv2d(c_flat, sd["cam_embed.weight"])
    pixel_flat = pixel.reshape(1, 1, 3, -1)
    cam = I_inv @ pixel_flat
    cam = F.pad(cam, (0, 0, 0, 1, 0, 0, 0, 0), value=1)
    d = E_inv @ cam
    d_flat = d.reshape(b * n, 4, h, w)
    d_embed = F.conv2d(d_flat, sd["img_embed.weight"])
    img_embed = d_embed - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    world = grid[:2].to(x.dtype)
    w_embed = F.conv2d(world[None], sd["bev_embed.weight"], sd["bev_embed.bias"])
    bev_embed = w_embed - c_embed
    bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
    query_pos = bev_embed.reshape(b, n, *bev_embed.shape[1:])
    feature_flat = feature.reshape(b * n, feat_dim, h, w)
    if cfg["no_image_features"]:
        key_flat = img_embed
    else:
        key_flat = img_embed + _bn_relu_conv(feature_flat, sd, "feature_proj")
    val_flat = _bn_relu_conv(feature_flat, sd, "feature_linear")
    query = query_pos + x[:, None]
    key = key_flat.reshape(b, n, *key_flat.shape[1:])
    val = val_flat.reshape(b, n, *val_flat.shape[1:])
    return cross_attention(query, key, val, x if cfg["skip"] else None, sd, cfg["heads"], cfg["dim_head"])


# ---- seeded parameters / inputs (numpy legacy stream, as the other oracles) ----

def make_config(heads=4, dim_head=32, image=512):
    return {"image_width": image, "image_height": image, "no_image_features": False, "heads": heads, "dim_head": dim_head,
            "qkv_bias": True, "skip": True}


def random_state_dict(feat_dim: int, dim: int, cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    rs = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    sd: Dict[str, Tensor] = {}

    def bn(p, c):
        sd[f"{p}.weight"] = t(1 + 0.1 * rs.standard_normal(c)); sd[f"{p}.bias"] = t(0.1 * rs.standard_normal(c))
        sd[f"{p}.running_mean"] = t(0.2 * rs.standard_normal(c)); sd[f"{p}.running_var"] = t(rs.uniform(0.5, 1.5, c))

    def lin(p, o, i, bias=True, shape=None):
        bnd = 1.0 / math.sqrt(i)
        sd[f"{p}.weight"] = t(rs.uniform(-bnd, bnd, shape or (o, i)))
        if bias:
            sd[f"{p}.bias"] = t(rs.uniform(-bnd, bnd, o))

    def ln(p, c):
        sd[f"{p}.weight"] = t(1 + 0.1 * rs.standard_normal(c)); sd[f"{p}.bias"] = t(0.1 * rs.standard_normal(c))

    for p in ("feature_linear", "feature_proj"):
        bn(f"{p}.0", feat_dim)
        lin(f"{p}.2", dim, feat_dim, bias=False, shape=(dim, feat_dim, 1, 1))
    lin("bev_embed", dim, 2, shape=(dim, 2, 1, 1))
    lin("img_embed", dim, 4, bias=False, shape=(dim, 4, 1, 1))
    lin("cam_embed", dim, 4, bias=False, shape=(dim, 4, 1, 1))
    hd = cfg["heads"] * cfg["dim_head"]
    for nme in ("to_q", "to_k", "to_v"):
        ln(f"cross_attend.{nme}.0", dim)
        lin(f"cross_attend.{nme}.1", hd, dim, bias=cfg["qkv_bias"])
    lin("cross_attend.proj", dim, hd)
    ln("cross_attend.prenorm", dim)
    lin("cross_attend.mlp.0", 2 * dim, dim)
    lin("cross_attend.mlp.2", dim, 2 * dim)
    ln("cross_attend.postnorm", dim)
    return sd


def synthetic_inputs(b, n, feat_dim, h, w, dim, H, W, seed=0, image=512):
    """Camera features + plausible pinhole intrinsics (f = 335.64 as in the reference's test yaml, rescaled) and four
    cameras looking forward / left / right / back from slightly different mounting points."""
    rs = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    x = t(0.5 * rs.standard_normal((b, dim, H, W)))
    feature = t(rs.standard_normal((b, n, feat_dim, h, w)))
    f = 335.64 * image / 800.0
    K = np.array([[f, 0, image / 2], [0, f, image / 2], [0, 0, 1]], np.float64)
    I_inv = np.tile(np.linalg.inv(K)[None, None], (b, n, 1, 1))
    E_inv = np.zeros((b, n, 4, 4))
    for bi in range(b):
        for ci in range(n):
            yaw = ci * math.pi / 2 + 0.05 * rs.standard_normal()
            c, s = math.cos(yaw), math.sin(yaw)
            # camera axes (x right, y down, z forward) expressed in the ego frame (x forward, y left, z up)
            R = np.array([[s, 0, c], [-c, 0, s], [0, -1, 0]], np.float64)
            T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [1.5 * c + 0.1 * rs.standard_normal(), 1.5 * s, 1.6 + 0.05 * bi]
            E_inv[bi, ci] = T
    return x, feature, t(I_inv), t(E_inv)
