"""Camera -> BEV lift of the reference's CVT encoder backed by libhmvit (HIP, gfx950): mirrors of ``BEVEmbedding``,
``CrossAttention`` and ``CrossViewAttention`` (``opencood/models/sub_modules/cvt_modules.py:43-91, 95-280``) with the same
constructors, ``forward`` signatures and ``state_dict`` names.  Eval mode only (BatchNorm running statistics), no CPU path.

Scope note: this is the cross-view attention itself.  The image backbone (``ResnetEncoder``, torchvision
arithmetic) and the ``ResNetBottleNeck`` refinement layers of ``CrossViewModule`` are not part of this module; the camera
slot of ``BevformerPointPillarHetero`` still takes any encoder honouring the slot contract.

Kernels: ``hmvit_cvt_embed`` (ray / BEV positional embeddings), ``hmvit_bn_relu_tokens`` (BN + ReLU + layout),
``hmvit_cross_attention`` (joint softmax over all cameras' keys); LayerNorm and every Linear / 1x1 convolution run on the
library's LayerNorm and GEMM kernels (f32; the attention core and its q / k / v projections also in f16 on the matrix cores).
"""
from __future__ import annotations

import contextlib
import ctypes

import torch
from torch import nn

from . import _lib

_F32 = _lib.PREC_F32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def generate_grid(height: int, width: int) -> torch.Tensor:
    """cvt_modules.py:15-26 (meshgrid((xs, ys)) with the default 'ij' indexing, exactly as written there)."""
    xs = torch.linspace(0, 1, width)
    ys = torch.linspace(0, 1, height)
    yy, xx = torch.meshgrid((xs, ys), indexing="ij")
    indices = torch.stack([xx, yy], 0)
    indices = torch.nn.functional.pad(indices, (0, 0, 0, 0, 0, 1), value=1)
    return indices[None]


class BEVEmbedding(nn.Module):
    """cvt_modules.py:43-91: learned BEV prior + the ego-frame coordinates of the query cells."""

    def __init__(self, dim, sigma, bev_height, bev_width, h_meters, w_meters, offset, decoder_blocks):
        super().__init__()
        h = bev_height // (2 ** len(decoder_blocks))
        w = bev_width // (2 ** len(decoder_blocks))
        grid = generate_grid(h, w).squeeze(0)
        grid[0] = bev_width * grid[0]
        grid[1] = bev_height * grid[1]
        sh, sw = bev_height / h_meters, bev_width / w_meters
        V = torch.tensor([[0.0, -sw, bev_width / 2.0], [-sh, 0.0, bev_height * offset + bev_height / 2.0], [0.0, 0.0, 1.0]])
        grid = (V.inverse() @ grid.reshape(3, -1)).reshape(3, h, w)
        self.register_buffer("grid", grid, persistent=False)
        self.learned_features = nn.Parameter(sigma * torch.randn(dim, h, w))

    def get_prior(self):
        return self.learned_features


def _layernorm(x2d, ln: nn.LayerNorm, prec=_F32):
    M, C = x2d.shape
    y = torch.empty(M, C, device=x2d.device, dtype=torch.float32 if prec == _F32 else torch.float16)
    _lib.check(_lib.lib.hmvit_layernorm(x2d.data_ptr(), y.data_ptr(), _lib.i32_array([0]), ln.weight.data_ptr(),
                                        ln.bias.data_ptr(), 1, M, C, prec, _stream()), "layernorm")
    return y


def _f16_weight(weight, N, K):
    """The f16 copy of a Linear / 1x1-convolution weight, made once per parameter value: kept as an attribute of the parameter
    object itself (a dictionary keyed by tensors would compare them elementwise) and refreshed when its storage, device or
    in-place version counter changes (optimiser step, load_state_dict, .to())."""
    hit = getattr(weight, "_hmvit_f16_copy", None)
    if hit is not None and hit[0] == weight.data_ptr() and hit[1] == weight._version and hit[2].device == weight.device:
        return hit[2]
    w2 = weight.detach().reshape(N, K).to(torch.float16).contiguous()
    weight._hmvit_f16_copy = (weight.data_ptr(), weight._version, w2)
    return w2


def _linear_f16(x2d, weight, bias=None, residual=None, gelu=False, out_f32=False):
    """f16 operands (an f32 input is converted first), f32 accumulate, f16 or f32 result; residual (f32) needs out_f32."""
    M, K = x2d.shape
    N = weight.shape[0]
    if x2d.dtype != torch.float16:
        x2d = x2d.to(torch.float16)
    y = torch.empty(M, N, device=x2d.device, dtype=torch.float32 if out_f32 else torch.float16)
    w2 = _f16_weight(weight, N, K)
    b2 = None if bias is None else bias.detach().float().contiguous()
    _lib.check(_lib.lib.hmvit_linear(x2d.data_ptr(), w2.data_ptr(), _ptr(b2), _ptr(residual), y.data_ptr(), M, N, K,
                                     1 if gelu else 0, 1 if out_f32 else 0, _lib.PREC_F16, _stream()), "linear")
    return y


_SPLIT_LINEARS = [False]


@contextlib.contextmanager
def split_linears(enabled: bool):
    """Inside the block the f32 Linears of this module family (``_linear``) run their products on split-f16 operands
    (HMVIT_PREC_SPLIT: f32 in, f32 out, fp32-class accuracy, the f16 matrix pipes) instead of the exact-f32 MFMA; set by the
    camera encoders for ``precision="split"``."""
    old = _SPLIT_LINEARS[0]
    _SPLIT_LINEARS[0] = bool(enabled)
    try:
        yield
    finally:
        _SPLIT_LINEARS[0] = old


def _linear(x2d, weight, bias=None, residual=None, gelu=False):
    M, K = x2d.shape
    N = weight.shape[0]
    y = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    w2 = weight.reshape(N, K).contiguous()
    prec = _lib.PREC_SPLIT if (_SPLIT_LINEARS[0] and K % 64 == 0) else _F32
    _lib.check(_lib.lib.hmvit_linear(x2d.data_ptr(), w2.data_ptr(), _ptr(bias), _ptr(residual), y.data_ptr(), M, N, K,
                                     1 if gelu else 0, 1, prec, _stream()), "linear")
    return y


class CrossAttention(nn.Module):
    """cvt_modules.py:95-173.  ``forward`` takes token-major tensors here: q (b, n, Q, dim), k (b, n, K, dim), v (b, n K, dim),
    skip (b, Q, dim) or None, and returns (b, Q, dim).  ``precision`` (attribute, not a constructor argument of the reference):
    "f32" keeps every product in f32; "f16" runs LayerNorm -> q / k / v projections -> attention with f16 operands on the
    matrix cores (f32 accumulate and softmax) when Q and K are multiples of 64."""

    precision = "f32"

    def __init__(self, dim, heads, dim_head, qkv_bias, norm=nn.LayerNorm):
        super().__init__()
        if dim_head != 32:
            raise NotImplementedError("cross attention kernel: dim_head must be 32")
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_k = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_v = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.proj = nn.Linear(heads * dim_head, dim)
        self.prenorm = norm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.postnorm = norm(dim)

    def _split_core_in_range(self) -> bool:
        """The split-product attention core takes q / k / v as (hi, lo) f16 halves at their own scale (no range normalisation in
        that kernel): it is used only while a bound on |q|, |k|, |v| - LayerNorm output (<= sqrt(dim) max|gamma| + max|beta| per
        element) through the projection (largest row L1 norm, + bias) - stays far inside f16's range; beyond it (weights hundreds
        of times their usual size) the exact-f32 kernel runs, as it did before round 4.  Cached per weight version."""
        params = [p_ for seq in (self.to_q, self.to_k, self.to_v) for p_ in seq.parameters()]
        key = tuple((p_.data_ptr(), p_._version) for p_ in params)
        if getattr(self, "_range_key", None) != key:
            worst = 0.0
            for seq in (self.to_q, self.to_k, self.to_v):
                ln, lin = seq[0], seq[1]
                dim = lin.weight.shape[1]
                x_max = float(dim) ** 0.5 * float(ln.weight.detach().abs().max()) + float(ln.bias.detach().abs().max())
                y_max = float(lin.weight.detach().abs().sum(1).max()) * x_max + (float(lin.bias.detach().abs().max()) if lin.bias is not None else 0.0)
                worst = max(worst, y_max)
            self._range_ok, self._range_key = bool(worst < 3.0e4), key      # f16 max 65504; non-finite weights fail the test too
        return self._range_ok

    def forward(self, q, k, v, skip=None):
        b, n, Q, dim = q.shape
        K = k.shape[2]
        hd = self.heads * self.dim_head
        if self.precision not in ("f32", "f16"):
            raise ValueError(f"CrossAttention.precision: {self.precision!r}")
        half = self.precision == "f16" and Q % 64 == 0 and K % 64 == 0 and dim % 64 == 0
        if half:
            F16 = _lib.PREC_F16
            qp = _linear_f16(_layernorm(q.reshape(-1, dim), self.to_q[0], F16), self.to_q[1].weight, self.to_q[1].bias)
            kp = _linear_f16(_layernorm(k.reshape(-1, dim), self.to_k[0], F16), self.to_k[1].weight, self.to_k[1].bias)
            vp = _linear_f16(_layernorm(v.reshape(-1, dim), self.to_v[0], F16), self.to_v[1].weight, self.to_v[1].bias)
        else:
            qp = _linear(_layernorm(q.reshape(-1, dim), self.to_q[0]), self.to_q[1].weight, self.to_q[1].bias)
            kp = _linear(_layernorm(k.reshape(-1, dim), self.to_k[0]), self.to_k[1].weight, self.to_k[1].bias)
            vp = _linear(_layernorm(v.reshape(-1, dim), self.to_v[0]), self.to_v[1].weight, self.to_v[1].bias)
        a = torch.empty(b * Q, hd, device=q.device, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_cross_attention(qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), a.data_ptr(), b, n, Q, K,
                                                  self.heads, self.dim_head,
                                                  _lib.PREC_F16 if half else (_lib.PREC_SPLIT if (_SPLIT_LINEARS[0] and self._split_core_in_range()) else _F32), _stream()),
                   "cross_attention")      # split model: the attention core on split-f16 products too (the library keeps the exact-f32 kernel for ragged sizes)
        res = None if skip is None else skip.reshape(-1, dim).contiguous()
        if half and dim % 64 == 0:
            # the rest of the block on f16 operands too (f32 accumulate; LayerNorm and both residual adds stay in f32)
            z = _linear_f16(a, self.proj.weight, self.proj.bias, residual=res, out_f32=True)
            z = _layernorm(z, self.prenorm)
            hdn = _linear_f16(z, self.mlp[0].weight, self.mlp[0].bias, gelu=True)
            z = _linear_f16(hdn, self.mlp[2].weight, self.mlp[2].bias, residual=z, out_f32=True)
        else:
            z = _linear(a, self.proj.weight, self.proj.bias, residual=res)
            z = _layernorm(z, self.prenorm)
            hdn = _linear(z, self.mlp[0].weight, self.mlp[0].bias, gelu=True)
            z = _linear(hdn, self.mlp[2].weight, self.mlp[2].bias, residual=z)
        z = _layernorm(z, self.postnorm)
        return z.reshape(b, Q, dim)


class CrossViewAttention(nn.Module):
    """cvt_modules.py:176-280; ``forward(x, bev, feature, I_inv, E_inv)`` -> (b, dim, H, W)."""

    def __init__(self, feat_height, feat_width, feat_dim, dim, config):
        super().__init__()
        if feat_height != feat_width:
            raise NotImplementedError("cross view attention: square feature maps only (generate_grid's axis order)")
        self.image_width, self.image_height = config["image_width"], config["image_height"]
        self.feat_height, self.feat_width = feat_height, feat_width
        self.feature_linear = nn.Sequential(nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        self.feature_proj = None if config["no_image_features"] else nn.Sequential(
            nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        self.bev_embed = nn.Conv2d(2, dim, 1)
        self.img_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.cam_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.cross_attend = CrossAttention(dim, config["heads"], config["dim_head"], config["qkv_bias"])
        self.skip = config["skip"]
        self.dim = dim

    @staticmethod
    def _bn_affine(bn: nn.BatchNorm2d):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()

    def _bn_relu_conv(self, seq, feature_flat, residual=None):
        bn_, C, h, w = feature_flat.shape
        scale, shift = self._bn_affine(seq[0])
        tok = torch.empty(bn_, h * w, C, device=feature_flat.device, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_bn_relu_tokens(feature_flat.data_ptr(), scale.data_ptr(), shift.data_ptr(), tok.data_ptr(),
                                                 bn_, C, h * w, _stream()), "bn_relu_tokens")
        if self.cross_attend.precision == "f16" and C % 64 == 0:
            return _linear_f16(tok.reshape(-1, C), seq[2].weight, None, residual=residual, out_f32=True)
        return _linear(tok.reshape(-1, C), seq[2].weight, None, residual=residual)

    def forward(self, x, bev, feature, I_inv, E_inv):
        if self.training:
            from .camera_train import cross_view_attention_forward
            return cross_view_attention_forward(self, x, bev, feature, I_inv, E_inv)
        if not x.is_cuda:
            raise RuntimeError("hm-vit_amd has no CPU path: pass CUDA tensors")
        b, n, feat_dim, h, w = feature.shape
        _, dim, H, W = x.shape
        x = x.contiguous().float()
        feature = feature.contiguous().float()
        I_inv = I_inv.reshape(b * n, 3, 3).contiguous().float()
        E_inv = E_inv.reshape(b * n, 4, 4).contiguous().float()
        grid = bev.grid.contiguous().float()
        dev = x.device
        lib = _lib.lib
        # positional embeddings
        key_pos = torch.empty(b * n, h * w, dim, device=dev, dtype=torch.float32)
        _lib.check(lib.hmvit_cvt_embed(0, I_inv.data_ptr(), E_inv.data_ptr(), None, self.img_embed.weight.data_ptr(), None,
                                       self.cam_embed.weight.data_ptr(), None, key_pos.data_ptr(), b, n, h, w, dim,
                                       float(self.image_width), float(self.image_height), _stream()), "cvt_embed")
        query = torch.empty(b * n, H * W, dim, device=dev, dtype=torch.float32)
        _lib.check(lib.hmvit_cvt_embed(1, None, E_inv.data_ptr(), grid.data_ptr(), self.bev_embed.weight.data_ptr(),
                                       self.bev_embed.bias.data_ptr(), self.cam_embed.weight.data_ptr(), x.data_ptr(),
                                       query.data_ptr(), b, n, H, W, dim, 0.0, 0.0, _stream()), "cvt_embed")
        feature_flat = feature.reshape(b * n, feat_dim, h, w)
        if self.feature_proj is not None:
            key = self._bn_relu_conv(self.feature_proj, feature_flat, residual=key_pos.reshape(-1, dim))
        else:
            key = key_pos.reshape(-1, dim)
        val = self._bn_relu_conv(self.feature_linear, feature_flat)
        skip = None
        if self.skip:
            skip = torch.empty(b, H * W, dim, device=dev, dtype=torch.float32)
            _lib.check(lib.hmvit_nchw_to_tokens(x.data_ptr(), skip.data_ptr(), b, dim, H * W, _stream()), "nchw_to_tokens")
        z = self.cross_attend(query.reshape(b, n, H * W, dim), key.reshape(b, n, h * w, dim), val.reshape(b, n * h * w, dim), skip)
        out = torch.empty(b, dim, H, W, device=dev, dtype=torch.float32)
        _lib.check(lib.hmvit_tokens_to_nchw(z.contiguous().data_ptr(), out.data_ptr(), b, dim, H * W, _stream()), "tokens_to_nchw")
        return out
