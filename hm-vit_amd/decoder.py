"""Drop-in ``HeteroDecoder`` (detection tail) backed by libhmvit's implicit-GEMM convolution.

Mirror of ``opencood/models/sub_modules/hetero_decoder.py:7-89`` (+ ``naive_decoder.py:7-92``): same params
dict, ``forward(x, mode, use_upsample=False) -> (psm, rm)``, same ``state_dict`` keys.  Only the
``use_upsample=False`` form the HM-ViT model uses (``bevformer_point_pillar_hetero.py:125``) is built.
Eval mode only (BatchNorm folded); no CPU path.
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib

_PREC = {"f32": _lib.PREC_F32, "f16": _lib.PREC_F16, "split": _lib.PREC_SPLIT}   # split: f32 maps, convolutions on split-f16 MFMA


class NaiveDecoder(nn.Module):
    def __init__(self, params: dict):
        super().__init__()
        self.num_ch_dec, self.num_layer, self.input_dim = params["num_ch_dec"], params["num_layer"], params["input_dim"]
        layers = []
        for i in range(self.num_layer - 1, -1, -1):
            cin = self.input_dim if i == self.num_layer - 1 else self.num_ch_dec[i + 1]
            cout = self.num_ch_dec[i]
            layers += [nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(True),
                       nn.Conv2d(cout, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(True)]
        self.decoder = nn.ModuleList(layers)


class NaiveCompressor(nn.Module):
    """Channel compressor for the exchanged BEV maps (``opencood/models/sub_modules/naive_compress.py:5-28``; the model's
    ``compression`` option, ``bevformer_point_pillar_hetero.py:69-71,116-117``): conv3x3(C -> C/r) + BN(eps 1e-3) + ReLU, then
    conv3x3(C/r -> C) + BN + ReLU, conv3x3(C -> C) + BN + ReLU.  Same constructor and ``state_dict`` names; eval mode
    (BatchNorm folded into the convolutions); ``forward(x (N, C, H, W)) -> (N, C, H, W)`` on the implicit-GEMM kernel."""

    def __init__(self, input_dim: int, compress_raito: int, precision: str = "split"):
        super().__init__()
        mid = input_dim // compress_raito
        self.encoder = nn.Sequential(nn.Conv2d(input_dim, mid, 3, 1, 1), nn.BatchNorm2d(mid, eps=1e-3, momentum=0.01), nn.ReLU())
        self.decoder = nn.Sequential(nn.Conv2d(mid, input_dim, 3, 1, 1), nn.BatchNorm2d(input_dim, eps=1e-3, momentum=0.01), nn.ReLU(),
                                     nn.Conv2d(input_dim, input_dim, 3, 1, 1), nn.BatchNorm2d(input_dim, eps=1e-3, momentum=0.01),
                                     nn.ReLU())
        self.input_dim = input_dim
        self.precision = precision
        self._prep = None

    def forward(self, x):
        from .camera import _Conv, _Prepared, _to_nchw, _to_nhwc      # shared convolution plumbing (channel padding, BN fold)
        if x.device.type != "cuda":
            raise RuntimeError("hm-vit_amd runs on the GPU only (HIP kernels, no CPU fallback)")
        if self.training:
            # batch-statistics BatchNorm + gradients: the training kernels of the detection tail (hm-vit_amd/tail_train.py)
            from . import tail_train as TT
            if (self.input_dim // 1) % 32 or self.encoder[0].out_channels % 32:
                raise NotImplementedError("NaiveCompressor training: channel counts must be multiples of 32 (compress ratio <= 8)")
            t = x.float().permute(0, 2, 3, 1).contiguous()
            for conv, bn in ((self.encoder[0], self.encoder[1]), (self.decoder[0], self.decoder[1]), (self.decoder[3], self.decoder[4])):
                t = TT.bn_relu_module(TT.Conv3x3.apply(t, conv.weight, conv.bias), bn)
            return t.permute(0, 3, 1, 2).contiguous()
        if self._prep is None:
            self._prep = _Prepared()
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        convs = self._prep.get(self, prec, lambda: [_Conv(self.encoder[0], self.encoder[1], prec, dt),
                                                    _Conv(self.decoder[0], self.decoder[1], prec, dt),
                                                    _Conv(self.decoder[3], self.decoder[4], prec, dt)])
        with torch.cuda.device(x.device):
            t = _to_nhwc(x.detach().float(), convs[0].cin, dt)
            for c in convs:
                t = c(t, relu=True)
            return _to_nchw(t, self.input_dim)


class HeteroDecoder(nn.Module):
    def __init__(self, params: dict, precision: str = "split"):
        super().__init__()
        dim = params["num_ch_dec"][0]
        self.camera_decoder = NaiveDecoder(params)
        self.lidar_decoder = NaiveDecoder(params)
        self.camera_cls_head = nn.Conv2d(dim, params["anchor_number"], kernel_size=1)
        self.camera_reg_head = nn.Conv2d(dim, 7 * params["anchor_number"], kernel_size=1)
        self.lidar_cls_head = nn.Conv2d(dim, params["anchor_number"], kernel_size=1)
        self.lidar_reg_head = nn.Conv2d(dim, 7 * params["anchor_number"], kernel_size=1)
        self.precision = precision
        self._prep, self._prep_key = None, None

    def _prepare(self, device, prec):
        tensors = list(self.parameters()) + list(self.buffers())
        key = (prec, str(device)) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key == self._prep_key:
            return self._prep
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32

        def conv(c, bn=None):
            w, b = c.weight.detach().float(), c.bias.detach().float()
            if bn is not None:
                s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                w, b = w * s[:, None, None, None], (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
            wmax = 0.0
            if prec == _lib.PREC_SPLIT:
                w, wmax = _lib.prescale_weights(w)              # exact power-of-two multiple, undone in the kernel's epilogue
            rows = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dt).contiguous()
            return dict(w=rows, b=b.contiguous(), cin=w.shape[1], cout=w.shape[0], k=w.shape[2], pad=c.padding[0], wmax=wmax,
                        img=_lib.conv_image(rows, w.shape[0], w.shape[1], w.shape[2], 1, c.padding[0], prec, wmax))

        prep = {}
        for t, name in ((0, "camera"), (1, "lidar")):
            dec = getattr(self, f"{name}_decoder").decoder
            prep[t] = {"convs": [conv(dec[i], dec[i + 1]) for i in range(0, len(dec), 3)],
                       "cls": conv(getattr(self, f"{name}_cls_head")), "reg": conv(getattr(self, f"{name}_reg_head"))}
        self._prep, self._prep_key = prep, key
        return prep

    def _forward_training(self, x, mode, torch_modules: bool = False):
        """Training mode (``train_camera.py:163-199``): BatchNorm works on batch statistics and updates its running buffers,
        and the tail needs gradients.  The layers are applied in the reference's order (``hetero_decoder.py:55-89``,
        ``naive_decoder.py:80-92`` with ``use_upsample=False``) through hm-vit_amd/tail_train.py: every convolution, the
        batch-statistics reductions, the normalisation and all their adjoints are libhmvit kernels behind
        ``torch.autograd.Function``s (f32 maps, split-f16 products); the parameters stay the reference's ``nn.Module``
        parameters.  ``torch_modules=True`` applies the modules themselves instead (the tests' reference for this path).
        ``.eval()`` switches back to the folded inference kernels."""
        from . import tail_train as TT
        ego = [int(v) for v in (mode[:, 0].tolist() if mode.device.type == "cpu" else mode[:, 0].cpu().tolist())]
        for v in ego:
            if v not in (0, 1):
                raise ValueError(f"Mode but be either 1 or 0 but received {v}")
        if x.device.type != "cuda":
            raise RuntimeError("hm-vit_amd runs on the GPU only (HIP kernels, no CPU fallback)")
        psm, rm = [None] * len(ego), [None] * len(ego)
        for v, name in ((0, "camera"), (1, "lidar")):
            idx = [b for b, e in enumerate(ego) if e == v]
            if not idx:
                continue
            layers = getattr(self, f"{name}_decoder").decoder
            cls_head, reg_head = getattr(self, f"{name}_cls_head"), getattr(self, f"{name}_reg_head")
            t = x[idx, 0]                                  # (n, C, H, W): all egos of this type share the BatchNorm batch
            if torch_modules:
                for layer in layers:
                    t = layer(t)
                p, r = cls_head(t), reg_head(t)
            else:
                t = t.float().permute(0, 2, 3, 1).contiguous()            # NHWC
                for k in range(0, len(layers), 3):                          # [Conv2d 3x3, BatchNorm2d, ReLU] blocks
                    conv, bn = layers[k], layers[k + 1]
                    t = TT.bn_relu_module(TT.Conv3x3.apply(t, conv.weight, conv.bias), bn)
                p = TT.Conv1x1.apply(t, cls_head.weight, cls_head.bias).permute(0, 3, 1, 2)
                r = TT.Conv1x1.apply(t, reg_head.weight, reg_head.bias).permute(0, 3, 1, 2)
            for k, b in enumerate(idx):
                psm[b], rm[b] = p[k], r[k]
        return torch.stack(psm, dim=0), torch.stack(rm, dim=0)

    def forward(self, x, mode, use_upsample=True):
        if use_upsample:
            raise NotImplementedError("HeteroDecoder: only use_upsample=False (the HM-ViT path) is built")
        if x.device.type != "cuda":
            raise RuntimeError("hm-vit_amd runs on the GPU only (HIP kernels, no CPU fallback)")
        if self.training:
            return self._forward_training(x, mode)
        B, L1, C, H, W = x.shape
        ego = [int(v) for v in (mode[:, 0].tolist() if mode.device.type == "cpu" else mode[:, 0].cpu().tolist())]
        for v in ego:
            if v not in (0, 1):
                raise ValueError(f"Mode but be either 1 or 0 but received {v}")
        dev = x.device
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        prep = self._prepare(dev, prec)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        A = self.camera_cls_head.out_channels
        psm = torch.empty(B, A, H, W, device=dev, dtype=torch.float32)
        rm = torch.empty(B, 7 * A, H, W, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            for t in (0, 1):
                idx = [b for b in range(B) if ego[b] == t]
                if not idx:
                    continue
                n = len(idx)
                xin = x[idx, 0].detach().float().contiguous()                    # (n, C, H, W)
                tok = torch.empty(n, H, W, C, device=dev, dtype=torch.float32)
                _lib.check(_lib.lib.hmvit_nchw_to_tokens(xin.data_ptr(), tok.data_ptr(), n, C, H * W, stream), "nchw_to_tokens")
                cur = tok.to(dt)
                for layer in prep[t]["convs"]:
                    y = torch.empty(n, H, W, layer["cout"], device=dev, dtype=dt)
                    if prec == _lib.PREC_SPLIT:
                        _lib.conv_range(cur, layer["wmax"], y, stream)
                    _lib.use_conv_image(layer.get("img"))
                    _lib.check(_lib.lib.hmvit_conv2d(cur.data_ptr(), layer["w"].data_ptr(), layer["b"].data_ptr(),
                                                     y.data_ptr(), n, H, W, layer["cin"], layer["cout"], layer["k"], 1,
                                                     layer["pad"], 1, layer["cout"], 0, 0, 0, prec, stream), "conv2d")
                    cur = y
                for head, dst in (("cls", psm), ("reg", rm)):
                    layer = prep[t][head]
                    y = torch.empty(n, H, W, layer["cout"], device=dev, dtype=torch.float32)
                    if prec == _lib.PREC_SPLIT:
                        _lib.conv_range(cur, layer["wmax"], y, stream)
                    _lib.check(_lib.lib.hmvit_conv2d(cur.data_ptr(), layer["w"].data_ptr(), layer["b"].data_ptr(),
                                                     y.data_ptr(), n, H, W, layer["cin"], layer["cout"], 1, 1, 0, 0,
                                                     layer["cout"], 0, 0, 1, prec, stream), "conv2d(head)")
                    out = torch.empty(n, layer["cout"], H, W, device=dev, dtype=torch.float32)
                    _lib.check(_lib.lib.hmvit_tokens_to_nchw(y.data_ptr(), out.data_ptr(), n, layer["cout"], H * W, stream),
                               "tokens_to_nchw")
                    dst[idx] = out
        return psm, rm
