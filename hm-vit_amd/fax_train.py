"""Training-mode forward of the FAX camera branch (``FaxCameraEncoder`` = ResnetEncoder -> FAXModule -> up-sampling NaiveDecoder,
fax_fused_transformer.py:12-57) on the autograd tape: the reference trains this branch through torch.autograd when the camera
backbone is not frozen (train_camera.py:109-120).  Every product with parameters runs through the autograd Functions of
``camera_train`` / ``tail_train`` (HIP forward and backward: Linear, LayerNorm, GELU, the windowed cross attention with its saved
log-sum-exp, 3 x 3 / 1 x 1 convolutions, BatchNorm on batch statistics); window / dilated-grid partitions, PixelUnshuffle, the
camera-geometry embeddings over 2-4 channels and residual adds are torch views and elementwise ops.

  * ``CrossViewSwapAttention`` (fax_modules.py:325-445): local-to-local and local-to-global ``CrossWinAttention`` (:205-252) with
    one "agent" per window for ``hmvit_cross_attention_train``, each followed by its pre-norm MLP, then ``postnorm``.
  * ``FAXModule`` level loop (:499-525): Bottlenecks, the down-sampling block (conv3x3 - PixelUnshuffle - conv3x3 - BN - ReLU -
    conv1x1 - BN, :478-492), and the closing self-attention with a relative-position bias (``Attention``, :136-180) on
    ``hmvit_attention_bias_train`` / ``_backward`` (round 4; rounds 2-3 ran its two products as library GEMMs): the kernel
    returns the gradient of the gathered (heads, N, N) bias, the table receives it through the index gather's autograd.
Checked against the oracle restatement under float64 autograd (tests/test_hip_camera_train.py)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import camera_train as CT
from . import tail_train as TT
from .cvt import generate_grid


def _win_tokens(t, w1, w2, grid=False):
    """(b, n, H, W, d) -> (b, X, Y, n, w1, w2, d): contiguous windows '(x w1) (y w2)' or the dilated grid '(w1 x) (w2 y)'."""
    b, n, H, W, d = t.shape
    if grid:
        return t.reshape(b, n, w1, H // w1, w2, W // w2, d).permute(0, 3, 5, 1, 2, 4, 6)
    return t.reshape(b, n, H // w1, w1, W // w2, w2, d).permute(0, 2, 4, 1, 3, 5, 6)


def cross_win_attention_forward(att, q_tok, k_tok, v_tok, skip_tok, q_win, f_win, grid: bool):
    """``CrossWinAttention.forward``: q_tok (b, nq, H, W, dim), k_tok / v_tok (b, n, h, w, dim), skip_tok (b, H, W, dim) or None
    -> (b, H, W, dim).  Inside window l every query of every query camera attends to the keys of all cameras in window l; the
    per-camera results are averaged (:246)."""
    b, nq, H, W, dim = q_tok.shape
    _, n, h, w, _ = k_tok.shape
    (W1, W2), (w1, w2) = q_win, f_win
    X, Y = H // W1, W // W2
    if X * Y != (h // w1) * (w // w2):
        raise ValueError(f"FAX: {X}x{Y} query windows but {h // w1}x{w // w2} feature windows")
    hd = att.heads * att.dim_head
    proj = lambda seq, t: CT.linear(CT.layer_norm(t.reshape(-1, dim), seq[0]), seq[1])
    qp = proj(att.to_q, q_tok).reshape(b, nq, H, W, hd)
    kp = proj(att.to_k, k_tok).reshape(b, n, h, w, hd)
    vp = proj(att.to_v, v_tok).reshape(b, n, h, w, hd)
    Q, K = nq * W1 * W2, n * w1 * w2
    qw = _win_tokens(qp, W1, W2).reshape(b * X * Y, 1, Q, hd)
    kw = _win_tokens(kp, w1, w2, grid).reshape(b * X * Y, 1, K, hd)
    vw = _win_tokens(vp, w1, w2, grid).reshape(b * X * Y, K, hd)
    a = CT.CrossAttnFn.apply(qw, kw, vw, att.heads, att.dim_head)                   # (b X Y, Q, hd)
    z = CT.linear(a.reshape(-1, hd), att.proj).reshape(b, X, Y, nq, W1, W2, dim).mean(3)
    z = z.permute(0, 1, 3, 2, 4, 5).reshape(b, H, W, dim)                           # reverse the window partition
    return z + skip_tok if skip_tok is not None else z


def _mlp(t, prenorm, mlp):
    b, H, W, dim = t.shape
    t2 = t.reshape(-1, dim)
    z = CT.layer_norm(t2, prenorm)
    return (t2 + CT.linear(CT.GeluFn.apply(CT.linear(z, mlp[0])), mlp[2])).reshape(b, H, W, dim)


def cross_view_swap_attention_forward(m, index, x, bev, feature, I_inv, E_inv):
    """``CrossViewSwapAttention.forward`` in training mode: x (b, dim, H, W), feature (b, n, feat_dim, h, w) -> (b, dim, H, W)."""
    b, n, feat_dim, h, w = feature.shape
    _, dim, H, W = x.shape
    dev = x.device
    x = x.float()
    I_inv = I_inv.reshape(b, n, 3, 3).float()
    E_inv = E_inv.reshape(b, n, 4, 4).float()
    pixel = generate_grid(h, w)[None].to(dev)                       # 1 1 3 h w
    pixel = pixel * torch.tensor([m.image_width, m.image_height, 1.0], device=dev).view(1, 1, 3, 1, 1)
    c = E_inv[..., -1:].reshape(b * n, 4, 1, 1)                     # camera centres
    c_embed = CT._small_conv(c, m.cam_embed)
    cam = CT._small_matmul(I_inv, pixel.reshape(1, 1, 3, h * w))    # pixel rays (geometry, no parameters)
    cam = F.pad(cam, (0, 0, 0, 1), value=1.0)
    d = CT._small_matmul(E_inv, cam).reshape(b * n, 4, h, w)
    img_embed = CT._small_conv(d, m.img_embed) - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    if m.bev_embed_flag:
        grid = getattr(bev, "grid%d" % index)[:2][None].to(dev).float()
        bev_embed = CT._small_conv(grid, m.bev_embed) - c_embed
        bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
        query = bev_embed.reshape(b, n, dim, H, W) + x[:, None]
    else:
        query = x[:, None]                                           # a single camera of queries (:393)
    feat = feature.reshape(b * n, feat_dim, h, w).permute(0, 2, 3, 1).contiguous().float()       # NHWC
    key = img_embed.permute(0, 2, 3, 1)                              # key before value, as fax_modules.py evaluates them
    if m.feature_proj is not None:
        key = key + CT.conv1x1(TT.bn_relu_module(feat, m.feature_proj[0]), m.feature_proj[2])
    val = CT.conv1x1(TT.bn_relu_module(feat, m.feature_linear[0]), m.feature_linear[2])
    key, val = key.reshape(b, n, h, w, dim), val.reshape(b, n, h, w, dim)
    w1, w2 = m.feat_win_size
    if h % w1 or w % w2:                                             # pad_divisble (:317-323)
        ph = ((h + w1) // w1) * w1 - h if h % w1 else 0
        pw = ((w + w2) // w2) * w2 - w if w % w2 else 0
        key, val = F.pad(key, (0, 0, 0, pw, 0, ph)), F.pad(val, (0, 0, 0, pw, 0, ph))
    q_tok = query.permute(0, 1, 3, 4, 2)                             # (b, nq, H, W, dim)
    x_tok = x.permute(0, 2, 3, 1)
    q1 = cross_win_attention_forward(m.cross_win_attend_1, q_tok, key, val, x_tok if m.skip else None, m.q_win_size, m.feat_win_size, False)
    q1 = _mlp(q1, m.prenorm_1, m.mlp_1)
    # local-to-global: the n repeated query copies of the reference give n identical results whose mean is that result
    q2 = cross_win_attention_forward(m.cross_win_attend_2, q1[:, None], key, val, q1 if m.skip else None, m.q_win_size, m.feat_win_size, True)
    q2 = _mlp(q2, m.prenorm_2, m.mlp_2)
    q2 = CT.layer_norm(q2.reshape(-1, dim), m.postnorm).reshape(b, H, W, dim)
    return q2.permute(0, 3, 1, 2)


def self_attention_forward(att, x):
    """``Attention.forward`` (fax_modules.py:136-180) in training mode: x (b, dim, h, w) -> (b, dim, h, w)."""
    b, dim, h, w = x.shape
    N, m, dh = h * w, att.heads, att.dim_head
    if N != att.rel_pos_indices.shape[0]:
        raise ValueError(f"Attention: map {h}x{w} does not match window_size {att.window_size}")
    tok = x.permute(0, 2, 3, 1).reshape(b * N, dim)
    qkv = CT.LinearFn.apply(tok, att.to_qkv.weight, None).reshape(b, N, 3, m, dh)
    if dh != 32:
        raise ValueError(f"Attention: dim_head {dh} (the HIP attention kernels are built for 32)")
    q, k, v = (qkv[:, :, i].reshape(b, N, m * dh) for i in range(3))               # token-major, head h in channels 32 h .. 32 h + 31
    bias = att.rel_pos_bias.weight[att.rel_pos_indices].permute(2, 0, 1)            # (m, N, N), gradient by indexing
    out = CT.AttnBiasFn.apply(q, k, v, bias, m, dh).reshape(b * N, dim)             # softmax(q k^T / sqrt(dh) + bias) v on libhmvit
    z = CT.LinearFn.apply(out, att.to_out[0].weight, None)
    z = att.to_out[1](z)                                                             # nn.Dropout (active in train())
    return z.reshape(b, h, w, dim).permute(0, 3, 1, 2)


def downsample_forward(seq, t):
    """FAXModule.downsample_layers[i][0] on an NHWC map: conv3x3 (no bias) - PixelUnshuffle(2) - conv3x3 - BN - ReLU - conv1x1 - BN."""
    y = CT.conv3x3(t, seq[0])
    n, H, W, c = y.shape
    # PixelUnshuffle(2) on NHWC: channel 4 c + 2 i + j of pixel (y, x) <- channel c of pixel (2 y + i, 2 x + j)
    y = y.reshape(n, H // 2, 2, W // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, H // 2, W // 2, c * 4).contiguous()
    y = TT.bn_relu_module(CT.conv3x3(y, seq[2]), seq[3])
    return TT.bn_relu_module(CT.conv1x1(y, seq[5]), seq[6], relu=False)


def fax_module_forward(fm, batch):
    """``FAXModule.forward`` in training mode -> (b, l, dim[-1], H, W)."""
    b, l, n = batch["camera"].shape[:3]
    I_inv = torch.linalg.inv_ex(batch["intrinsic"].reshape(b * l, n, 3, 3).float())[0]   # (inv_ex: no host read of the status)
    E_inv = batch["extrinsic"].reshape(b * l, n, 4, 4).float()
    x = fm.bev_embedding.get_prior()[None].expand(b * l, -1, -1, -1)
    n_levels = len(fm.cross_views)
    for i, (cross_view, feature) in enumerate(zip(fm.cross_views, batch["features"])):
        feature = feature.reshape(b * l, n, *feature.shape[3:])
        x = cross_view_swap_attention_forward(cross_view, i, x, fm.bev_embedding, feature, I_inv, E_inv)
        if len(fm.layers[i]) or i < n_levels - 1:
            t = x.permute(0, 2, 3, 1).contiguous()
            for blk in fm.layers[i]:
                t = CT.bottleneck_block(blk, t)
            if i < n_levels - 1:
                t = downsample_forward(fm.downsample_layers[i][0], t)
            x = t.permute(0, 3, 1, 2)
    x = self_attention_forward(fm.self_attn, x)
    return x.reshape(b, l, *x.shape[1:])


def fax_camera_encoder_forward(enc, batch_camera):
    """``FaxCameraEncoder.forward`` in training mode: images -> ResNet pyramid -> FAX lift -> decoder -> (N, num_ch_dec[0], Hb, Wb)."""
    cam = batch_camera["camera"]
    feats = CT.resnet_encoder_forward(enc.encoder, cam[:, None])
    x = fax_module_forward(enc.fax, {"camera": cam[:, None], "intrinsic": batch_camera["intrinsic"][:, None],
                                     "extrinsic": batch_camera["extrinsic"][:, None], "features": feats})[:, 0]
    return CT.naive_decoder_forward(enc.decoder, x, use_upsample=True)
