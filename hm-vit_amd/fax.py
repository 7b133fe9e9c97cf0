"""FAX camera -> BEV lift backed by libhmvit (HIP, gfx950): mirrors of the reference's ``CrossWinAttention``,
``CrossViewSwapAttention``, ``Attention``, ``BEVEmbedding`` and ``FAXModule`` (``opencood/models/sub_modules/fax_modules.py:43-525``)
and of the camera branch ``FaxFusedTransformer`` assembles (``opencood/models/fax_fused_transformer.py:12-64``:
``ResnetEncoder`` -> ``FAXModule`` -> up-sampling ``NaiveDecoder``), with the same constructor dicts, ``state_dict`` names and
the encoder-slot contract of the HM-ViT model (``set_return_features()``, ``forward(batch_camera) -> (N, 256, H, W)``,
SURVEY 8f-4).  No CPU path; train() mode runs hm-vit_amd/fax_train.py (HIP forward + backward on the autograd tape).

Kernels: the positional embeddings (``hmvit_cvt_embed``), BatchNorm + ReLU + layout (``hmvit_bn_relu_tokens``), LayerNorm and
every Linear / 1x1 convolution (``hmvit_layernorm`` / ``hmvit_linear``), the windowed cross attention on
``hmvit_cross_attention`` (one "agent" per window: inside window l every query of every camera attends to the keys of all
cameras in window l, fax_modules.py:205-252), the closing self-attention with its relative-position bias on
``hmvit_attention_bias``, every 3x3 convolution / Bottleneck on the implicit-GEMM kernel.  Window and dilated-grid partitions
are index permutations of the token tensors (torch views + one copy each): layout plumbing, as in the reference's rearranges.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .camera import Bottleneck, ResnetEncoder, _Conv, _Prepared, _to_nchw, _to_nhwc, _PREC
from . import cvt as _cvt
from .cvt import _layernorm, _linear, _linear_f16, _stream, generate_grid
from .decoder import NaiveDecoder


class BEVEmbedding(nn.Module):
    """fax_modules.py:43-93: learned BEV prior at the first level's size + ego-frame coordinates of the cells of every level."""

    def __init__(self, dim, sigma, bev_height, bev_width, h_meters, w_meters, offset, upsample_scales):
        super().__init__()
        sh, sw = bev_height / h_meters, bev_width / w_meters
        V = torch.tensor([[0.0, -sw, bev_width / 2.0], [-sh, 0.0, bev_height * offset + bev_height / 2.0], [0.0, 0.0, 1.0]])
        for i, scale in enumerate(upsample_scales):
            h, w = bev_height // scale, bev_width // scale
            grid = generate_grid(h, w).squeeze(0)
            grid[0] = bev_width * grid[0]
            grid[1] = bev_height * grid[1]
            self.register_buffer("grid%d" % i, (V.inverse() @ grid.reshape(3, -1)).reshape(3, h, w), persistent=False)
        self.learned_features = nn.Parameter(sigma * torch.randn(dim, bev_height // upsample_scales[0], bev_width // upsample_scales[0]))

    def get_prior(self):
        return self.learned_features


class CrossWinAttention(nn.Module):
    """fax_modules.py:183-252 (parameters; the arithmetic is driven by CrossViewSwapAttention below)."""

    def __init__(self, dim, heads, dim_head, qkv_bias, rel_pos_emb=False, norm=nn.LayerNorm):
        super().__init__()
        if dim_head != 32:
            raise NotImplementedError("cross attention kernel: dim_head must be 32")
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_k = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_v = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.proj = nn.Linear(heads * dim_head, dim)


def _win_tokens(t, w1, w2, grid=False):
    """(b, n, H, W, d) -> (b, X, Y, n, w1, w2, d): contiguous windows '(x w1) (y w2)' or the dilated grid '(w1 x) (w2 y)'."""
    b, n, H, W, d = t.shape
    if grid:
        return t.reshape(b, n, w1, H // w1, w2, W // w2, d).permute(0, 3, 5, 1, 2, 4, 6)
    return t.reshape(b, n, H // w1, w1, W // w2, w2, d).permute(0, 2, 4, 1, 3, 5, 6)


class CrossViewSwapAttention(nn.Module):
    """fax_modules.py:255-445; ``forward(index, x, bev, feature, I_inv, E_inv)`` -> (b, dim, H, W).  ``precision`` "f16" runs
    the projections and the attention cores on f16 operands (f32 accumulate / softmax / LayerNorm / residuals)."""

    precision = "f32"

    def __init__(self, feat_height, feat_width, feat_dim, dim, index, image_height, image_width, qkv_bias, q_win_size,
                 feat_win_size, heads, dim_head, bev_embedding_flag, rel_pos_emb=False, no_image_features=False, skip=True,
                 norm=nn.LayerNorm):
        super().__init__()
        if feat_height != feat_width:
            raise NotImplementedError("cross view attention: square feature maps only (generate_grid's axis order)")
        self.image_width, self.image_height = image_width, image_height
        self.feature_linear = nn.Sequential(nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        self.feature_proj = None if no_image_features else nn.Sequential(
            nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        self.bev_embed_flag = bev_embedding_flag[index]
        if self.bev_embed_flag:
            self.bev_embed = nn.Conv2d(2, dim, 1)
        self.img_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.cam_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.q_win_size, self.feat_win_size = q_win_size[index], feat_win_size[index]
        self.cross_win_attend_1 = CrossWinAttention(dim, heads[index], dim_head[index], qkv_bias)
        self.cross_win_attend_2 = CrossWinAttention(dim, heads[index], dim_head[index], qkv_bias)
        self.skip = skip
        self.prenorm_1, self.prenorm_2 = norm(dim), norm(dim)
        self.mlp_1 = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.mlp_2 = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.postnorm = norm(dim)
        self.dim = dim

    # ---- helpers ----
    def _half(self, *sizes):
        return self.precision == "f16" and all(s % 64 == 0 for s in sizes)

    def _bn_relu_conv(self, seq, feature_flat, residual=None):
        bn_, C, h, w = feature_flat.shape
        scale = (seq[0].weight / torch.sqrt(seq[0].running_var + seq[0].eps)).contiguous()
        shift = (seq[0].bias - seq[0].running_mean * scale).contiguous()
        tok = torch.empty(bn_, h * w, C, device=feature_flat.device, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_bn_relu_tokens(feature_flat.data_ptr(), scale.data_ptr(), shift.data_ptr(), tok.data_ptr(),
                                                 bn_, C, h * w, _stream()), "bn_relu_tokens")
        if self._half(C):
            return _linear_f16(tok.reshape(-1, C), seq[2].weight, None, residual=residual, out_f32=True)
        return _linear(tok.reshape(-1, C), seq[2].weight, None, residual=residual)

    def _project(self, seq, tok2d, half):
        """LayerNorm + Linear of one of to_q / to_k / to_v on a (M, dim) token matrix."""
        if half:
            return _linear_f16(_layernorm(tok2d, seq[0], _lib.PREC_F16), seq[1].weight, seq[1].bias)
        return _linear(_layernorm(tok2d, seq[0]), seq[1].weight, seq[1].bias)

    def _window_attention(self, att: CrossWinAttention, q_tok, k_tok, v_tok, skip_tok, grid: bool):
        """q_tok (b, nq, H, W, dim), k_tok / v_tok (b, n, h, w, dim) f32 tokens, skip_tok (b, H, W, dim) or None
        -> (b, H, W, dim): CrossWinAttention.forward with the window (or, for the keys, dilated grid) partition."""
        b, nq, H, W, dim = q_tok.shape
        _, n, h, w, _ = k_tok.shape
        (W1, W2), (w1, w2) = self.q_win_size, self.feat_win_size
        X, Y = H // W1, W // W2
        if X * Y != (h // w1) * (w // w2):
            raise ValueError(f"FAX: {X}x{Y} query windows but {h // w1}x{w // w2} feature windows")
        hd = att.heads * att.dim_head
        Q, K = nq * W1 * W2, n * w1 * w2
        half = self._half(Q, K, dim)
        # per-token projections in the natural layout, then the partition of the projected rows
        qp = self._project(att.to_q, q_tok.reshape(-1, dim), half).reshape(b, nq, H, W, hd)
        kp = self._project(att.to_k, k_tok.reshape(-1, dim), half).reshape(b, n, h, w, hd)
        vp = self._project(att.to_v, v_tok.reshape(-1, dim), half).reshape(b, n, h, w, hd)
        qw = _win_tokens(qp, W1, W2).contiguous()                       # (b, X, Y, nq, W1, W2, hd)
        kw = _win_tokens(kp, w1, w2, grid).contiguous()
        vw = _win_tokens(vp, w1, w2, grid).contiguous()
        a = torch.empty(b * X * Y * Q, hd, device=q_tok.device, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_cross_attention(qw.data_ptr(), kw.data_ptr(), vw.data_ptr(), a.data_ptr(), b * X * Y, 1, Q, K,
                                                  att.heads, att.dim_head,
                                                  _lib.PREC_F16 if half else (_lib.PREC_SPLIT if (_cvt._SPLIT_LINEARS[0] and _cvt.CrossAttention._split_core_in_range(att)) else _lib.PREC_F32), _stream()),
                   "cross_attention")      # split model: split-f16 products where the window sizes allow (cvt.CrossAttention does the same)
        z = _linear_f16(a, att.proj.weight, att.proj.bias, out_f32=True) if half else _linear(a, att.proj.weight, att.proj.bias)
        z = z.reshape(b, X, Y, nq, W1, W2, dim).mean(3)                  # reduce the query cameras (fax_modules.py:246)
        z = z.permute(0, 1, 3, 2, 4, 5).reshape(b, H, W, dim)            # reverse the window partition
        return z + skip_tok if skip_tok is not None else z

    def _mlp(self, t, prenorm, mlp):
        b, H, W, dim = t.shape
        t2 = t.reshape(-1, dim).contiguous()
        z = _layernorm(t2, prenorm)
        if self._half(dim):
            hdn = _linear_f16(z, mlp[0].weight, mlp[0].bias, gelu=True)
            return _linear_f16(hdn, mlp[2].weight, mlp[2].bias, residual=t2, out_f32=True).reshape(b, H, W, dim)
        hdn = _linear(z, mlp[0].weight, mlp[0].bias, gelu=True)
        return _linear(hdn, mlp[2].weight, mlp[2].bias, residual=t2).reshape(b, H, W, dim)

    def forward(self, index, x, bev, feature, I_inv, E_inv):
        if self.training:
            if not torch.is_grad_enabled():
                raise RuntimeError("hmvit_amd.CrossViewSwapAttention: train() mode under no_grad; call eval() for inference")
            from .fax_train import cross_view_swap_attention_forward
            return cross_view_swap_attention_forward(self, index, x, bev, feature, I_inv, E_inv)
        if not x.is_cuda:
            raise RuntimeError("hm-vit_amd has no CPU path: pass CUDA tensors")
        b, n, feat_dim, h, w = feature.shape
        _, dim, H, W = x.shape
        x = x.contiguous().float()
        feature = feature.contiguous().float()
        I_inv = I_inv.reshape(b * n, 3, 3).contiguous().float()
        E_inv = E_inv.reshape(b * n, 4, 4).contiguous().float()
        dev, lib = x.device, _lib.lib
        key_pos = torch.empty(b * n, h * w, dim, device=dev, dtype=torch.float32)
        _lib.check(lib.hmvit_cvt_embed(0, I_inv.data_ptr(), E_inv.data_ptr(), None, self.img_embed.weight.data_ptr(), None,
                                       self.cam_embed.weight.data_ptr(), None, key_pos.data_ptr(), b, n, h, w, dim,
                                       float(self.image_width), float(self.image_height), _stream()), "cvt_embed")
        x_tok = torch.empty(b, H * W, dim, device=dev, dtype=torch.float32)
        _lib.check(lib.hmvit_nchw_to_tokens(x.data_ptr(), x_tok.data_ptr(), b, dim, H * W, _stream()), "nchw_to_tokens")
        x_tok = x_tok.reshape(b, H, W, dim)
        if self.bev_embed_flag:
            grid = getattr(bev, "grid%d" % index).contiguous().float()
            query = torch.empty(b * n, H * W, dim, device=dev, dtype=torch.float32)
            _lib.check(lib.hmvit_cvt_embed(1, None, E_inv.data_ptr(), grid.data_ptr(), self.bev_embed.weight.data_ptr(),
                                           self.bev_embed.bias.data_ptr(), self.cam_embed.weight.data_ptr(), x.data_ptr(),
                                           query.data_ptr(), b, n, H, W, dim, 0.0, 0.0, _stream()), "cvt_embed")
            query = query.reshape(b, n, H, W, dim)
        else:
            query = x_tok[:, None]                                       # x[:, None]: a single camera of queries (:393)
        feature_flat = feature.reshape(b * n, feat_dim, h, w)
        key = (self._bn_relu_conv(self.feature_proj, feature_flat, residual=key_pos.reshape(-1, dim))
               if self.feature_proj is not None else key_pos.reshape(-1, dim)).reshape(b, n, h, w, dim)
        val = self._bn_relu_conv(self.feature_linear, feature_flat).reshape(b, n, h, w, dim)
        w1, w2 = self.feat_win_size
        if h % w1 or w % w2:                                             # pad_divisble (:317-323)
            ph = ((h + w1) // w1) * w1 - h if h % w1 else 0
            pw = ((w + w2) // w2) * w2 - w if w % w2 else 0
            key = torch.nn.functional.pad(key, (0, 0, 0, pw, 0, ph))
            val = torch.nn.functional.pad(val, (0, 0, 0, pw, 0, ph))
        skip = x_tok if self.skip else None
        q1 = self._window_attention(self.cross_win_attend_1, query, key, val, skip, grid=False)     # local-to-local
        q1 = self._mlp(q1, self.prenorm_1, self.mlp_1)
        # local-to-global: the n repeated query copies of the reference give n identical results whose mean is that result
        q2 = self._window_attention(self.cross_win_attend_2, q1[:, None], key, val, q1 if self.skip else None, grid=True)
        q2 = self._mlp(q2, self.prenorm_2, self.mlp_2)
        q2 = _layernorm(q2.reshape(-1, dim).contiguous(), self.postnorm).reshape(b, H * W, dim)
        out = torch.empty(b, dim, H, W, device=dev, dtype=torch.float32)
        _lib.check(lib.hmvit_tokens_to_nchw(q2.contiguous().data_ptr(), out.data_ptr(), b, dim, H * W, _stream()), "tokens_to_nchw")
        return out


class Attention(nn.Module):
    """fax_modules.py:96-180: self-attention over the whole (h, w) map with a relative-position bias; ``forward(x (b, d, h, w))``."""

    def __init__(self, dim, dim_head=32, dropout=0., window_size=25):
        super().__init__()
        assert dim % dim_head == 0, "dimension should be divisible by dimension per head"
        if dim_head != 32:
            raise NotImplementedError("attention kernel: dim_head must be 32")
        self.heads, self.dim_head, self.window_size = dim // dim_head, dim_head, window_size
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(dropout))
        self.rel_pos_bias = nn.Embedding((2 * window_size - 1) ** 2, self.heads)
        pos = torch.arange(window_size)
        grid = torch.stack(torch.meshgrid(pos, pos, indexing="ij")).reshape(2, -1).t()
        rel = grid[:, None] - grid[None, :] + window_size - 1
        self.register_buffer("rel_pos_indices", (rel * torch.tensor([2 * window_size - 1, 1])).sum(-1), persistent=False)

    def forward(self, x):
        if self.training:
            from .fax_train import self_attention_forward
            return self_attention_forward(self, x)
        b, dim, h, w = x.shape
        if h * w != self.rel_pos_indices.shape[0]:
            raise ValueError(f"Attention: map {h}x{w} does not match window_size {self.window_size}")
        dev = x.device
        tok = torch.empty(b, h * w, dim, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_nchw_to_tokens(x.contiguous().float().data_ptr(), tok.data_ptr(), b, dim, h * w, _stream()), "nchw_to_tokens")
        qkv = _linear(tok.reshape(-1, dim), self.to_qkv.weight).reshape(b, h * w, 3, dim)
        q, k, v = (qkv[:, :, i].contiguous() for i in range(3))
        bias = self.rel_pos_bias.weight.detach().float()[self.rel_pos_indices].permute(2, 0, 1).contiguous()     # (heads, N, N)
        a = torch.empty(b, h * w, dim, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_attention_bias(q.data_ptr(), k.data_ptr(), v.data_ptr(), bias.data_ptr(), a.data_ptr(), b, h * w, h * w,
                                                 self.heads, self.dim_head, _stream()), "attention_bias")
        z = _linear(a.reshape(-1, dim), self.to_out[0].weight)
        out = torch.empty(b, dim, h, w, device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_tokens_to_nchw(z.data_ptr(), out.data_ptr(), b, dim, h * w, _stream()), "tokens_to_nchw")
        return out


class FAXModule(nn.Module):
    """fax_modules.py:448-525; ``forward(batch)`` with 'camera' (b, l, n, ...) for its leading shape, 'intrinsic' (b, l, n, 3, 3),
    'extrinsic' (b, l, n, 4, 4), 'features': list of (b, l, n, C, h, w) -> (b, l, dim[-1], H, W)."""

    def __init__(self, config: dict, precision: str = "split"):
        super().__init__()
        middle, dim = config["middle"], config["dim"]
        self.backbone_output_shape = config["backbone_output_shape"]
        assert len(middle) == len(self.backbone_output_shape)
        cross_views, layers, downs = [], [], []
        for i, (feat_shape, num_layers) in enumerate(zip(self.backbone_output_shape, middle)):
            _, _, _, feat_dim, feat_height, feat_width = feat_shape
            cross_views.append(CrossViewSwapAttention(feat_height, feat_width, feat_dim, dim[i], i, **config["cross_view"],
                                                      **config["cross_view_swap"]))
            layers.append(nn.Sequential(*[Bottleneck(dim[i], dim[i] // 4) for _ in range(num_layers)]))
            if i < len(middle) - 1:
                downs.append(nn.Sequential(nn.Sequential(
                    nn.Conv2d(dim[i], dim[i] // 4, 3, 1, 1, bias=False), nn.PixelUnshuffle(2),
                    nn.Conv2d(dim[i + 1], dim[i + 1], 3, padding=1, bias=False), nn.BatchNorm2d(dim[i + 1]), nn.ReLU(inplace=True),
                    nn.Conv2d(dim[i + 1], dim[i + 1], 1, padding=0, bias=False), nn.BatchNorm2d(dim[i + 1]))))
        self.bev_embedding = BEVEmbedding(dim[0], **config["bev_embedding"])
        self.cross_views = nn.ModuleList(cross_views)
        self.layers = nn.ModuleList(layers)
        self.downsample_layers = nn.ModuleList(downs)
        self.self_attn = Attention(dim[-1], **config["self_attn"])
        self.dim = dim
        self.precision = precision
        self._prep = _Prepared()

    def _build(self, prec, dt):
        return {"layers": [[{"c1": _Conv(b.conv1, b.bn1, prec, dt), "c2": _Conv(b.conv2, b.bn2, prec, dt),
                             "c3": _Conv(b.conv3, b.bn3, prec, dt)} for b in layer] for layer in self.layers],
                "down": [{"a": _Conv(d[0][0], None, prec, dt), "b": _Conv(d[0][2], d[0][3], prec, dt),
                          "c": _Conv(d[0][5], d[0][6], prec, dt)} for d in self.downsample_layers]}

    def forward(self, batch):
        from .cvt import split_linears
        with split_linears(self.precision == "split"):      # f32 Linears on split-f16 operands in the fp32-parity fast mode
            return self._forward(batch)

    def _forward(self, batch):
        if self.training:
            from .fax_train import fax_module_forward
            return fax_module_forward(self, batch)
        b, l, n = batch["camera"].shape[:3]
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        prep = self._prep.get(self, prec, lambda: self._build(prec, dt))
        for cv in self.cross_views:
            cv.precision = "f32" if self.precision == "split" else self.precision
        I_inv = torch.linalg.inv_ex(batch["intrinsic"].reshape(b * l, n, 3, 3).float())[0]   # (inv_ex: no host read of the status)      # 3x3 inverses: host-side plumbing, as the reference
        E_inv = batch["extrinsic"].reshape(b * l, n, 4, 4).float()
        x = self.bev_embedding.get_prior().detach().float()[None].repeat(b * l, 1, 1, 1).contiguous()
        n_levels = len(self.cross_views)
        for i, (cross_view, feature) in enumerate(zip(self.cross_views, batch["features"])):
            feature = feature.reshape(b * l, n, *feature.shape[3:])
            x = cross_view(i, x, self.bev_embedding, feature, I_inv, E_inv)
            if prep["layers"][i] or i < n_levels - 1:
                cin = prep["layers"][i][0]["c1"].cin if prep["layers"][i] else prep["down"][i]["a"].cin
                t = _to_nhwc(x, cin, dt)
                for blk in prep["layers"][i]:
                    t = blk["c3"](blk["c2"](blk["c1"](t)), relu=True, residual=t)
                if i < n_levels - 1:
                    d = prep["down"][i]
                    y = d["a"](t, relu=False)[..., :self.dim[i] // 4]                    # conv3x3 (no bias, no activation)
                    nn_, H, W, c = y.shape
                    # PixelUnshuffle(2) on NHWC: channel 4 c + 2 i + j of pixel (y, x) <- channel c of pixel (2 y + i, 2 x + j)
                    y = y.reshape(nn_, H // 2, 2, W // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(nn_, H // 2, W // 2, c * 4)
                    if y.shape[-1] != d["b"].cin:
                        y = torch.nn.functional.pad(y, (0, d["b"].cin - y.shape[-1]))
                    t = d["c"](d["b"](y.contiguous(), relu=True), relu=False)
                    x = _to_nchw(t, self.dim[i + 1])
                else:
                    x = _to_nchw(t, self.dim[i])
        x = self.self_attn(x)
        return x.reshape(b, l, *x.shape[1:])


class FaxCameraEncoder(nn.Module):
    """The camera branch of ``FaxFusedTransformer`` (fax_fused_transformer.py:12-57) for HM-ViT's camera slot.  config:
    {'encoder': ResnetEncoder params, 'fax': FAXModule config, 'decoder': NaiveDecoder params}; ``forward(batch_camera)`` with
    'camera' (N, n_cam, H, W, 3), 'intrinsic' (N, n_cam, 3, 3), 'extrinsic' (N, n_cam, 4, 4) -> (N, num_ch_dec[0], Hb, Wb)."""

    def __init__(self, config: dict, precision: str = "split"):
        super().__init__()
        self.encoder = ResnetEncoder(config["encoder"], precision=precision)
        fax = dict(config["fax"])
        fax["backbone_output_shape"] = self.encoder.output_shapes
        self.fax = FAXModule(fax, precision=precision)
        self.decoder = NaiveDecoder(config["decoder"])
        self.cls_head = nn.Conv2d(256, config.get("anchor_number", 2), kernel_size=1)       # present in the reference module, unused
        self.reg_head = nn.Conv2d(256, 7 * config.get("anchor_number", 2), kernel_size=1)   # when it only returns features
        self.precision = precision
        self.return_features = False
        self._prep = _Prepared()

    def set_return_features(self):
        self.return_features = True

    def forward(self, batch_camera):
        if not self.return_features:
            raise NotImplementedError("FaxCameraEncoder serves the HM-ViT camera slot: call set_return_features() first")
        if self.training:
            # un-frozen camera backbone (train_camera.py:109-120): HIP forward + backward on the autograd tape
            if not torch.is_grad_enabled():
                raise RuntimeError("hmvit_amd.FaxCameraEncoder: train() mode under no_grad; call eval() for inference")
            from .fax_train import fax_camera_encoder_forward
            return fax_camera_encoder_forward(self, batch_camera)
        cam = batch_camera["camera"]
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        dec = self.decoder.decoder
        convs = self._prep.get(self.decoder, prec, lambda: [_Conv(dec[i], dec[i + 1], prec, dt) for i in range(0, len(dec), 3)])
        feats = self.encoder(cam[:, None])                                          # camera.unsqueeze(1): (N, 1, n, ...)
        x = self.fax({"camera": cam[:, None], "intrinsic": batch_camera["intrinsic"][:, None],
                      "extrinsic": batch_camera["extrinsic"][:, None], "features": feats})[:, 0]   # (N, dim, Hq, Wq)
        t = _to_nhwc(x, convs[0].cin, dt)
        for i in range(0, len(convs), 2):
            t = convs[i](t)
            t = convs[i + 1](t, up2=True)            # NaiveDecoder.upsample between the two convolutions of a layer
        return _to_nchw(t, self.decoder.num_ch_dec[0])
