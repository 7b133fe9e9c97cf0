"""Camera branch for HM-ViT's camera slot, backed by libhmvit (HIP, gfx950): mirrors of the reference's ``ResnetEncoder``
(``opencood/models/backbones/resnet_ms.py:8-89``), ``CrossViewModule`` (``opencood/models/sub_modules/cvt_modules.py:283-331``)
and the up-sampling ``NaiveDecoder`` (``naive_decoder.py:8-92``), assembled as ``CvtCameraEncoder`` the way
``FaxFusedTransformer`` assembles its camera branch (``fax_fused_transformer.py:37-57``): images -> ResNet pyramid ->
cross-view attention + bottlenecks per level -> decoder -> (N, C, Hb, Wb) BEV features, with the ``set_return_features()``
/ ``forward(batch_camera)`` contract of the model's encoder slot (``base_camera_lidar_intermediate.py:15-28``).

Same constructor dicts and ``state_dict`` names (the ResNet keeps torchvision's names, so ImageNet / reference checkpoints
load).  Eval mode only (BatchNorm folded into the convolutions), no CPU path.  Every convolution runs on the implicit-GEMM
kernel (``hmvit_conv2d_ex``: residual add of the ResNet blocks and the decoder's nearest x2 upsampling are operands of the
convolution, never separate passes); channel counts that the kernel's K slab does not divide (the 32-channel bottlenecks)
are zero-padded once at weight-preparation time, and the 3-channel 7x7 stem runs row-packed (``hmvit_conv2d_rowpack``).
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib
from .cvt import BEVEmbedding, CrossViewAttention
from .decoder import NaiveDecoder

_PREC = {"f32": _lib.PREC_F32, "f16": _lib.PREC_F16, "split": _lib.PREC_SPLIT}   # split: f32 maps, convolutions on split-f16 MFMA
_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}
_EXPANSION = {18: 1, 34: 1, 50: 4, 101: 4, 152: 4}      # BasicBlock / Bottleneck (resnet_ms.py:27-31)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cpad(c: int, prec: int) -> int:
    q = 64 if prec == _lib.PREC_F16 else 32
    return (c + q - 1) // q * q


class _Conv:
    """One Conv2d (+ eval BatchNorm) prepared for hmvit_conv2d_ex: weight (Cout_p, k*k*Cin_p) in the kernel's k order,
    channels zero-padded to the K-slab granule."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d | None, prec: int, dt):
        w = conv.weight.detach().float()
        b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        if bn is not None:
            s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            w, b = w * s[:, None, None, None], (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
        co, ci, k, _ = w.shape
        self.cin, self.cout, self.k = _cpad(ci, prec), _cpad(co, prec), k
        self.wmax = 0.0                      # range information of the split convolutions (hmvit_conv_range)
        if prec == _lib.PREC_SPLIT:
            w, self.wmax = _lib.prescale_weights(w)
        wp = torch.zeros(self.cout, k, k, self.cin, device=w.device)
        wp[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
        self.w = wp.reshape(self.cout, -1).to(dt).contiguous()
        self.b = torch.zeros(self.cout, device=w.device)
        self.b[:co] = b
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.prec, self.dt = prec, dt
        self.img = _lib.conv_image(self.w, self.cout, self.cin, k, self.stride, self.pad, prec, self.wmax)

    def __call__(self, x, relu=True, residual=None, up2=False):
        n, H, W, ci = x.shape
        assert ci == self.cin, (ci, self.cin)
        if up2:
            H, W = 2 * H, 2 * W
        Ho = (H + 2 * self.pad - self.k) // self.stride + 1
        Wo = (W + 2 * self.pad - self.k) // self.stride + 1
        y = torch.empty(n, Ho, Wo, self.cout, device=x.device, dtype=self.dt)
        if self.prec == _lib.PREC_SPLIT:
            _lib.conv_range(x, self.wmax, y, _stream())
        _lib.use_conv_image(self.img)
        _lib.check(_lib.lib.hmvit_conv2d_ex(x.data_ptr(), self.w.data_ptr(), self.b.data_ptr(),
                                            residual.data_ptr() if residual is not None else None, y.data_ptr(), n, H, W, self.cin,
                                            self.cout, self.k, self.stride, self.pad, 1 if relu else 0, 1 if up2 else 0, 0,
                                            self.prec, _stream()), "conv2d_ex")
        return y


class _StemConv:
    """The 7x7 / stride 2 / 3-channel stem on hmvit_conv2d_rowpack: the image is copied once into a zero-bordered 4-channel
    NHWC map and every kernel row becomes one contiguous run of 8 pixels x 4 channels of the GEMM's K axis (K = 256 with
    147 useful taps, instead of one 64-channel slab per tap)."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, prec: int, dt):
        w = conv.weight.detach().float()
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w, b = w * s[:, None, None, None], bn.bias.detach().float() - bn.running_mean.detach().float() * s
        co, ci, k, _ = w.shape
        assert ci <= 4 and k <= 8 and conv.bias is None
        self.cout, self.k, self.stride, self.pad = _cpad(co, prec), k, conv.stride[0], conv.padding[0]
        self.wmax = 0.0
        if prec == _lib.PREC_SPLIT:
            w, self.wmax = _lib.prescale_weights(w)
        wp = torch.zeros(self.cout, 8, 8, 4, device=w.device)
        wp[:co, :k, :k, :ci] = w.permute(0, 2, 3, 1)
        self.w = wp.reshape(self.cout, 256).to(dt).contiguous()
        self.b = torch.zeros(self.cout, device=w.device)
        self.b[:co] = b
        self.prec, self.dt = prec, dt

    def __call__(self, img_nhwc):
        n, H, W, c = img_nhwc.shape
        Ho = (H + 2 * self.pad - self.k) // self.stride + 1
        Wo = (W + 2 * self.pad - self.k) // self.stride + 1
        Hp, Wp = (Ho - 1) * self.stride + 8, (Wo - 1) * self.stride + 8          # rows / pixels the kernel touches
        Hp, Wp = max(Hp, H + self.pad), max(Wp, W + self.pad)
        xp = torch.zeros(n, Hp, Wp, 4, device=img_nhwc.device, dtype=self.dt)
        xp[:, self.pad:self.pad + H, self.pad:self.pad + W, :c] = img_nhwc
        y = torch.empty(n, Ho, Wo, self.cout, device=xp.device, dtype=self.dt)
        if self.prec == _lib.PREC_SPLIT:
            _lib.conv_range(xp, self.wmax, y, _stream())
        _lib.check(_lib.lib.hmvit_conv2d_rowpack(xp.data_ptr(), self.w.data_ptr(), self.b.data_ptr(), y.data_ptr(), n, Hp, Wp, Ho, Wo,
                                                 self.cout, 8, self.stride, 1, self.prec, _stream()), "conv2d_rowpack")
        return y


def _to_nhwc(x_nchw, c_pad, dt):
    """(n, C, H, W) f32 -> (n, H, W, c_pad) in dt (extra channels zero)."""
    n, C, H, W = x_nchw.shape
    tok = torch.empty(n, H, W, C, device=x_nchw.device, dtype=torch.float32)
    _lib.check(_lib.lib.hmvit_nchw_to_tokens(x_nchw.contiguous().data_ptr(), tok.data_ptr(), n, C, H * W, _stream()), "nchw_to_tokens")
    if c_pad != C:
        tok = torch.nn.functional.pad(tok, (0, c_pad - C))
    return tok.to(dt).contiguous()


def _to_nchw(x_nhwc, C):
    """(n, H, W, Cp) -> (n, C, H, W) f32."""
    n, H, W, Cp = x_nhwc.shape
    tok = x_nhwc[..., :C].float().contiguous()
    out = torch.empty(n, C, H, W, device=tok.device, dtype=torch.float32)
    _lib.check(_lib.lib.hmvit_tokens_to_nchw(tok.data_ptr(), out.data_ptr(), n, C, H * W, _stream()), "tokens_to_nchw")
    return out


class _Prepared:
    """Caches the folded / padded weights of a module per (precision, parameter versions)."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, module, prec, build):
        tensors = list(module.parameters()) + list(module.buffers())
        key = (prec,) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.key:
            self.key, self.val = key, build()
        return self.val


# ---- ResNet (torchvision naming) ----

class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


class _BottleneckBlock(nn.Module):
    """torchvision ``Bottleneck`` of the ResNet-50 / 101 / 152 trunks: 1x1 (-> planes) - 3x3 with the stride (the "v1.5"
    placement torchvision has used since 0.3) - 1x1 (-> 4 planes), identity or 1x1-strided downsample, ReLU after the sum."""

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))


class _ResNet(nn.Module):
    def __init__(self, num_layers):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (nb, co) in enumerate(zip(_BLOCKS[num_layers], (64, 128, 256, 512))):
            blocks = []
            for bi in range(nb):
                stride = 2 if (li > 0 and bi == 0) else 1
                if _EXPANSION[num_layers] == 1:
                    blocks.append(_BasicBlock(cin, co, stride))
                    cin = co
                else:
                    blocks.append(_BottleneckBlock(cin, co, stride))
                    cin = 4 * co
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))


class ResnetEncoder(nn.Module):
    def __init__(self, params: dict, precision: str = "split"):
        super().__init__()
        self.num_layers = params["num_layers"]
        if self.num_layers not in _BLOCKS:
            raise ValueError("{} is not a valid number of resnet layers".format(self.num_layers))       # resnet_ms.py:33-36
        self.idx_pick = params["id_pick"]
        self.encoder = _ResNet(self.num_layers)
        self.precision = precision
        self._prep = _Prepared()
        ih, iw = params["image_height"], params["image_width"]
        ex = _EXPANSION[self.num_layers]
        self.channels = [64 * ex, 128 * ex, 256 * ex, 512 * ex]
        self.output_shapes = [torch.Size([1, 1, 1, c, ih // s, iw // s]) for c, s in zip(self.channels, (4, 8, 16, 32))]
        if isinstance(self.idx_pick, list):
            self.output_shapes = [self.output_shapes[i] for i in self.idx_pick]

    def _build(self, prec, dt):
        e = self.encoder
        prep = {"stem": _StemConv(e.conv1, e.bn1, prec, dt), "layers": []}
        for li in range(4):
            blocks = []
            for blk in getattr(e, f"layer{li + 1}"):
                blocks.append({"c1": _Conv(blk.conv1, blk.bn1, prec, dt), "c2": _Conv(blk.conv2, blk.bn2, prec, dt),
                               "c3": _Conv(blk.conv3, blk.bn3, prec, dt) if hasattr(blk, "conv3") else None,
                               "down": _Conv(blk.downsample[0], blk.downsample[1], prec, dt) if blk.downsample is not None else None})
            prep["layers"].append(blocks)
        return prep

    def forward(self, input_images):
        if self.training:
            # batch-statistics BatchNorm + gradients: the training-mode layer functions (hm-vit_amd/camera_train.py)
            from .camera_train import resnet_encoder_forward
            return resnet_encoder_forward(self, input_images)
        if not input_images.is_cuda:
            raise RuntimeError("hm-vit_amd has no CPU path: pass CUDA tensors")
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        prep = self._prep.get(self, prec, lambda: self._build(prec, dt))
        b, l, m, h, w, c = input_images.shape
        x = prep["stem"](input_images.reshape(b * l * m, h, w, c))        # NHWC already
        n, H, W, C = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty(n, Ho, Wo, C, device=x.device, dtype=dt)
        _lib.check(_lib.lib.hmvit_maxpool2d(x.data_ptr(), y.data_ptr(), n, H, W, C, 3, 2, 1, prec, _stream()), "maxpool2d")
        _lib.inherit_range(y, x)                # a maximum over a post-ReLU map: the stem's bound holds
        x = y
        outs = []
        for li, blocks in enumerate(prep["layers"]):
            for blk in blocks:
                idt = blk["down"](x, relu=False) if blk["down"] is not None else x
                if blk["c3"] is None:
                    x = blk["c2"](blk["c1"](x), relu=True, residual=idt)
                else:
                    x = blk["c3"](blk["c2"](blk["c1"](x)), relu=True, residual=idt)
            c_real = self.channels[li]
            f = _to_nchw(x, c_real)
            outs.append(f.reshape(b, l, m, *f.shape[1:]))
        return [outs[i] for i in self.idx_pick] if isinstance(self.idx_pick, list) else outs[self.idx_pick]


# ---- cross view module ----

class Bottleneck(nn.Module):
    """torchvision Bottleneck(c, c // 4) as built by ``ResNetBottleNeck`` (cvt_modules.py:13): stride 1, no downsample."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)


class CrossViewModule(nn.Module):
    def __init__(self, config: dict, precision: str = "split"):
        super().__init__()
        middle, dim = config["middle"], config["dim"]
        self.backbone_output_shape = config["backbone_output_shape"]
        assert len(middle) == len(self.backbone_output_shape)
        cross_views, layers = [], []
        for feat_shape, num_layers in zip(self.backbone_output_shape, middle):
            _, _, _, feat_dim, feat_height, feat_width = feat_shape
            cross_views.append(CrossViewAttention(feat_height, feat_width, feat_dim, dim, config["cross_view"]))
            layers.append(nn.Sequential(*[Bottleneck(dim, dim // 4) for _ in range(num_layers)]))
        self.bev_embedding = BEVEmbedding(dim, **config["bev_embedding"])
        self.cross_views = nn.ModuleList(cross_views)
        self.layers = nn.ModuleList(layers)
        self.dim = dim
        self.precision = precision
        self._prep = _Prepared()

    def _build(self, prec, dt):
        return [[{"c1": _Conv(b.conv1, b.bn1, prec, dt), "c2": _Conv(b.conv2, b.bn2, prec, dt), "c3": _Conv(b.conv3, b.bn3, prec, dt)}
                 for b in layer] for layer in self.layers]

    def forward(self, batch):
        from .cvt import split_linears
        with split_linears(self.precision == "split"):      # f32 Linears on split-f16 operands in the fp32-parity fast mode
            return self._forward(batch)

    def _forward(self, batch):
        """batch: 'inputs' (b, l, n, ...) only for its leading shape, 'intrinsic' (b, l, n, 3, 3), 'extrinsic' (b, l, n, 4, 4),
        'features': list of (b, l, n, C, h, w).  Returns (b, l, dim, H, W)."""
        if self.training:
            from .camera_train import cross_view_module_forward
            return cross_view_module_forward(self, batch)
        b, l, n = batch["inputs"].shape[:3]
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        prep = self._prep.get(self, prec, lambda: self._build(prec, dt))
        for cross_view in self.cross_views:
            cross_view.cross_attend.precision = "f32" if self.precision == "split" else self.precision   # split: f32 attention, split-operand convolutions
        I_inv = torch.linalg.inv_ex(batch["intrinsic"].reshape(b * l, n, 3, 3).float())[0]   # (inv_ex: no host read of the status)      # 3x3 inverses: host-side plumbing, as the reference
        E_inv = batch["extrinsic"].reshape(b * l, n, 4, 4).float()
        x = self.bev_embedding.get_prior().detach().float()[None].repeat(b * l, 1, 1, 1).contiguous()
        for cross_view, feature, layer in zip(self.cross_views, batch["features"], prep):
            feature = feature.reshape(b * l, n, *feature.shape[3:])
            x = cross_view(x, self.bev_embedding, feature, I_inv, E_inv)
            if layer:
                t = _to_nhwc(x, layer[0]["c1"].cin, dt)
                for blk in layer:
                    t = blk["c3"](blk["c2"](blk["c1"](t)), relu=True, residual=t)
                x = _to_nchw(t, self.dim)
        return x.reshape(b, l, *x.shape[1:])


# ---- the assembled camera encoder ----

class CvtCameraEncoder(nn.Module):
    """config: {'encoder': ResnetEncoder params, 'cvm': CrossViewModule config, 'decoder': NaiveDecoder params}."""

    def __init__(self, config: dict, precision: str = "split"):
        super().__init__()
        self.encoder = ResnetEncoder(config["encoder"], precision=precision)
        self.cvm = CrossViewModule(config["cvm"], precision=precision)
        self.decoder = NaiveDecoder(config["decoder"])
        self.precision = precision
        self.return_features = False
        self._prep = _Prepared()

    def set_return_features(self):
        self.return_features = True

    def _build(self, prec, dt):
        dec = self.decoder.decoder
        return [_Conv(dec[i], dec[i + 1], prec, dt) for i in range(0, len(dec), 3)]

    def forward(self, batch_camera):
        if self.training:
            # train_camera.py without --fix_camera_backbone: ResNet, cross-view lift and decoder on the autograd tape
            from .camera_train import cvt_camera_encoder_forward
            return cvt_camera_encoder_forward(self, batch_camera)
        cam = batch_camera["camera"]
        n_agents = cam.shape[0]
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        convs = self._prep.get(self.decoder, prec, lambda: self._build(prec, dt))
        feats = self.encoder(cam[None])
        x = self.cvm({"inputs": cam[None], "intrinsic": batch_camera["intrinsic"][None], "extrinsic": batch_camera["extrinsic"][None],
                      "features": feats})[0]                                        # (N, dim, Hq, Wq)
        t = _to_nhwc(x, convs[0].cin, dt)
        for i in range(0, len(convs), 2):
            t = convs[i](t)
            t = convs[i + 1](t, up2=True)            # NaiveDecoder.upsample between the two convolutions of a layer
        return _to_nchw(t, self.decoder.num_ch_dec[0])
