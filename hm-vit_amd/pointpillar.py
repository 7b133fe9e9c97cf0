"""Drop-in ``PointPillar`` LiDAR BEV encoder backed by libhmvit (HIP, gfx950).

Mirror of ``opencood/models/point_pillar.py:9-62``: same constructor ``args`` dict, same
``set_return_features()`` / ``forward(data_dict)`` contract (``data_dict['processed_lidar']`` with
``voxel_features (Nv, 32, 4)``, ``voxel_coords (Nv, 4) [agent, z, y, x]``, ``voxel_num_points (Nv)``),
same ``state_dict`` key names, so ``self.lidar_encoder = PointPillar(config['lidar'])`` in
``bevformer_point_pillar_hetero.py:56`` works unchanged.  Eval mode only (BatchNorm running
statistics are folded into the convolutions); no CPU path.

Kernels (csrc/enc.hip): PFN + scatter in one HBM-bound pass, every Conv2d / ConvTranspose2d as an
MFMA implicit GEMM on NHWC maps; ``torch.cat`` of the three up-sampled maps is free (each deconv
writes its channel window of the 384-channel buffer).
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib

_PREC = {"f32": _lib.PREC_F32, "f16": _lib.PREC_F16, "split": _lib.PREC_SPLIT}   # split: f32 maps, convolutions on split-f16 MFMA


def _block(cin, cout, n_layers, stride):
    layers = [nn.ZeroPad2d(1), nn.Conv2d(cin, cout, 3, stride=stride, padding=0, bias=False),
              nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01), nn.ReLU()]
    for _ in range(n_layers):
        layers += [nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01),
                   nn.ReLU()]
    return nn.Sequential(*layers)


class _PFNLayer(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear = nn.Linear(cin, cout, bias=False)
        self.norm = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01)


class _PillarVFE(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if not (cfg["use_norm"] and cfg["use_absolute_xyz"]) or cfg["with_distance"] or list(cfg["num_filters"]) != [64]:
            raise NotImplementedError("pillar_vfe: only use_norm, use_absolute_xyz, no distance, num_filters [64]")
        self.pfn_layers = nn.ModuleList([_PFNLayer(10, 64)])


class _Backbone(nn.Module):
    def __init__(self, cfg, cin):
        super().__init__()
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        for n_layers, stride, cout, us, cu in zip(cfg["layer_nums"], cfg["layer_strides"], cfg["num_filters"],
                                                  cfg["upsample_strides"], cfg["num_upsample_filter"]):
            if us < 1:
                raise NotImplementedError("upsample_strides < 1")
            self.blocks.append(_block(cin, cout, n_layers, stride))
            self.deblocks.append(nn.Sequential(nn.ConvTranspose2d(cout, cu, us, stride=us, bias=False),
                                               nn.BatchNorm2d(cu, eps=1e-3, momentum=0.01), nn.ReLU()))
            cin = cout


class _DoubleConv(nn.Module):
    def __init__(self, cin, cout, k, stride, pad):
        super().__init__()
        self.double_conv = nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=pad), nn.ReLU(inplace=True),
                                         nn.Conv2d(cout, cout, 3, padding=1), nn.ReLU(inplace=True))


class _DownsampleConv(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList()
        cin = cfg["input_dim"]
        for k, dim, stride, pad in zip(cfg["kernal_size"], cfg["dim"], cfg["stride"], cfg["padding"]):
            self.layers.append(_DoubleConv(cin, dim, k, stride, pad))
            cin = dim


class PointPillar(nn.Module):
    def __init__(self, args: dict, precision: str = "split"):
        super().__init__()
        self.args = args
        self.pillar_vfe = _PillarVFE(args["pillar_vfe"])
        self.scatter_cfg = args["point_pillar_scatter"]
        self.backbone = _Backbone(args["base_bev_backbone"], 64)
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = _DownsampleConv(args["shrink_header"])
        self.cls_head = nn.Conv2d(args["cls_head_dim"], args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(args["cls_head_dim"], 7 * args["anchor_number"], kernel_size=1)
        self.return_features = False
        self.trace = None            # a list: forward appends (stage name, f32 copy of the stage's output) - error reports only
        self.precision = precision
        self._prep = None
        self._prep_key = None
        self.oob_count = None      # device counter of pillars dropped because their indices fall outside the canvas

    def set_return_features(self):
        self.return_features = True

    # ---- weight preparation: BatchNorm folding + implicit-GEMM layouts (cached per parameter version) ----
    @staticmethod
    def _fold_bn(bn):
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        return scale, bn.bias.detach().float() - bn.running_mean.detach().float() * scale

    def _prepare(self, device, prec):
        tensors = list(self.parameters()) + list(self.buffers())
        key = (prec, str(device)) + tuple((t.data_ptr(), t._version) for t in tensors)
        if key == self._prep_key:
            return self._prep
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32

        def conv(c, bn=None, pad=None):
            pad = c.padding[0] if pad is None else pad          # (a block's first convolution: ZeroPad2d(1) + padding 0)
            w = c.weight.detach().float()                       # (Cout, Cin, k, k)
            b = c.bias.detach().float() if c.bias is not None else torch.zeros(w.shape[0], device=w.device)
            if bn is not None:
                s, sh = self._fold_bn(bn)
                w, b = w * s[:, None, None, None], b * s + sh
            wmax = 0.0
            if prec == _lib.PREC_SPLIT:
                w, wmax = _lib.prescale_weights(w)              # exact power-of-two multiple, undone in the kernel's epilogue
            rows = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dt).contiguous()
            return dict(w=rows, b=b.contiguous(), cin=w.shape[1], cout=w.shape[0], k=w.shape[2], stride=c.stride[0],
                        pad=pad, wmax=wmax,
                        img=_lib.conv_image(rows, w.shape[0], w.shape[1], w.shape[2], c.stride[0], pad, prec, wmax))

        def deconv(c, bn):
            w = c.weight.detach().float()                       # (Cin, Cout, s, s)
            s, sh = self._fold_bn(bn)
            w = w * s[None, :, None, None]
            us = c.stride[0]
            wmax = 0.0
            if prec == _lib.PREC_SPLIT:
                w, wmax = _lib.prescale_weights(w)
            rows = w.permute(2, 3, 1, 0).reshape(us * us * w.shape[1], w.shape[0]).to(dt).contiguous()
            return dict(w=rows, b=sh.contiguous(), cin=w.shape[0], cout=w.shape[1], us=us, wmax=wmax,
                        img=_lib.conv_image(rows, rows.shape[0], w.shape[0], 1, 1, 0, prec, wmax, deconv=True))

        pfn = self.pillar_vfe.pfn_layers[0]
        s, sh = self._fold_bn(pfn.norm)
        prep = {"pfn_w": (pfn.linear.weight.detach().float() * s[:, None]).contiguous(), "pfn_shift": sh.contiguous(),
                "blocks": [], "deblocks": [], "shrink": []}
        for blk, de in zip(self.backbone.blocks, self.backbone.deblocks):
            layers = [conv(blk[1], blk[2], pad=1)]               # ZeroPad2d(1) + padding 0: one weight image, built for pad 1
            k = 4
            while k < len(blk):
                layers.append(conv(blk[k], blk[k + 1]))
                k += 3
            prep["blocks"].append(layers)
            prep["deblocks"].append(deconv(de[0], de[1]))
        if self.shrink_flag:
            for dc in self.shrink_conv.layers:
                prep["shrink"].append([conv(dc.double_conv[0]), conv(dc.double_conv[2])])
        prep["cls"], prep["reg"] = conv(self.cls_head), conv(self.reg_head)
        self._prep, self._prep_key = prep, key
        return prep

    # ---- launch helpers ----
    @staticmethod
    def _conv(x, layer, N, H, W, out, ctot, coff, relu, out_f32, prec, stream):
        Ho = (H + 2 * layer["pad"] - layer["k"]) // layer["stride"] + 1
        Wo = (W + 2 * layer["pad"] - layer["k"]) // layer["stride"] + 1
        if prec == _lib.PREC_SPLIT:
            _lib.conv_range(x, layer["wmax"], out, stream)       # max |x| from x's producer, max |out| for its consumers
        _lib.use_conv_image(layer.get("img"))
        _lib.check(_lib.lib.hmvit_conv2d(x.data_ptr(), layer["w"].data_ptr(), layer["b"].data_ptr(), out.data_ptr(), N, H,
                                         W, layer["cin"], layer["cout"], layer["k"], layer["stride"], layer["pad"],
                                         int(relu), ctot, coff, 0, int(out_f32), prec, stream), "hmvit_conv2d")
        return Ho, Wo

    def forward(self, data_dict):
        lidar = data_dict["processed_lidar"]
        vf, vc, vn = lidar["voxel_features"], lidar["voxel_coords"], lidar["voxel_num_points"]
        if vf.device.type != "cuda":
            raise RuntimeError("hm-vit_amd runs on the GPU only (HIP kernels, no CPU fallback)")
        if self.training:
            # batch-statistics BatchNorm and gradients: the training-mode layer functions (hm-vit_amd/encoder_train.py); the folded
            # inference kernels below serve .eval()
            from .encoder_train import pointpillar_train_forward
            return pointpillar_train_forward(self, data_dict)
        dev = vf.device
        prec = _PREC[self.precision]
        dt = torch.float16 if prec == _lib.PREC_F16 else torch.float32
        prep = self._prepare(dev, prec)
        nx, ny, nz = [int(v) for v in self.scatter_cfg["grid_size"]]
        assert nz == 1
        if vf.dim() != 3 or tuple(vf.shape[1:]) != (32, 4) or vc.dim() != 2 or vc.shape[1] != 4 or vn.shape[0] != vf.shape[0]:
            raise ValueError(f"voxel_features must be (Nv, 32, 4), voxel_coords (Nv, 4) [agent, z, y, x] and voxel_num_points "
                             f"(Nv,); got {tuple(vf.shape)}, {tuple(vc.shape)}, {tuple(vn.shape)}")
        vf = vf.detach().float().contiguous()
        vc = vc.detach().to(torch.int32).contiguous()
        vn = vn.detach().to(torch.int32).contiguous()
        # point_pillar_scatter.py:18 reads the agent count back from the coordinates (a device synchronisation per call); a
        # caller that knows it passes it along as batch['n_agents'] (the assembled model does)
        n_agents = int(data_dict["n_agents"]) if "n_agents" in data_dict else int(vc[:, 0].max().item()) + 1
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            canvas = torch.zeros(n_agents, ny, nx, 64, device=dev, dtype=dt)
            # pillars outside the canvas (bad agent index / coordinates) are dropped and counted here instead of raising like
            # the reference's indexed scatter (reading the counter is a device synchronisation: the caller's choice)
            if self.oob_count is None or self.oob_count.device != dev:
                self.oob_count = torch.zeros(1, dtype=torch.int32, device=dev)
            vs = (ctypes.c_float * 3)(*[float(v) for v in self.args["voxel_size"]])
            rng = (ctypes.c_float * 6)(*[float(v) for v in self.args["lidar_range"]])
            if prec == _lib.PREC_SPLIT:
                _lib.announce_output_range(canvas)               # max |canvas| comes out of the scatter kernel (no extra pass)
            _lib.check(_lib.lib.hmvit_pfn_scatter(vf.data_ptr(), vc.data_ptr(), vn.data_ptr(), prep["pfn_w"].data_ptr(),
                                                  prep["pfn_shift"].data_ptr(), canvas.data_ptr(), None, vf.shape[0], nx,
                                                  ny, n_agents, self.oob_count.data_ptr(), vs, rng, prec, stream),
                       "hmvit_pfn_scatter")
            x, H, W = canvas, ny, nx
            note = (lambda name, t: self.trace.append((name, t.float().clone()))) if self.trace is not None else (lambda *a: None)
            note("pfn+scatter", canvas)
            cat = None
            ctot = sum(d["cout"] for d in prep["deblocks"])
            coff = 0
            for layers, de in zip(prep["blocks"], prep["deblocks"]):
                for layer in layers:
                    Ho = (H + 2 * layer["pad"] - layer["k"]) // layer["stride"] + 1
                    Wo = (W + 2 * layer["pad"] - layer["k"]) // layer["stride"] + 1
                    y = torch.empty(n_agents, Ho, Wo, layer["cout"], device=dev, dtype=dt)
                    self._conv(x, layer, n_agents, H, W, y, layer["cout"], 0, True, False, prec, stream)
                    x, H, W = y, Ho, Wo
                note(f"block{len(layers)}x{layers[-1]['cout']}", x)
                us = de["us"]
                if cat is None:
                    cat = torch.empty(n_agents, H * us, W * us, ctot, device=dev, dtype=dt)
                    Hc, Wc = H * us, W * us
                assert (H * us, W * us) == (Hc, Wc), "up-sampled maps must share one size"
                if prec == _lib.PREC_SPLIT:
                    _lib.conv_range(x, de["wmax"], cat, stream, share_out=True)
                _lib.use_conv_image(de.get("img"))
                _lib.check(_lib.lib.hmvit_conv2d(x.data_ptr(), de["w"].data_ptr(), de["b"].data_ptr(), cat.data_ptr(),
                                                 n_agents, H, W, de["cin"], de["cout"], 1, 1, 0, 1, ctot, coff, us, 0, prec,
                                                 stream), "hmvit_conv2d(deconv)")
                coff += de["cout"]
            x, H, W, C = cat, Hc, Wc, ctot
            note("deblocks(concat)", cat)
            convs = [layer for pair in prep["shrink"] for layer in pair]
            tail = convs if self.return_features else convs   # heads read the shrunk map
            for i, layer in enumerate(convs):
                last = self.return_features and i == len(convs) - 1
                Ho = (H + 2 * layer["pad"] - layer["k"]) // layer["stride"] + 1
                Wo = (W + 2 * layer["pad"] - layer["k"]) // layer["stride"] + 1
                y = torch.empty(n_agents, Ho, Wo, layer["cout"], device=dev, dtype=torch.float32 if last else dt)
                self._conv(x, layer, n_agents, H, W, y, layer["cout"], 0, True, last, prec, stream)
                x, H, W, C = y, Ho, Wo, layer["cout"]
                note(f"shrink{i}", x)

            def to_nchw(t32, ch):
                out = torch.empty(n_agents, ch, H, W, device=dev, dtype=torch.float32)
                _lib.check(_lib.lib.hmvit_tokens_to_nchw(t32.data_ptr(), out.data_ptr(), n_agents, ch, H * W, stream),
                           "hmvit_tokens_to_nchw")
                return out

            if self.return_features:
                if x.dtype != torch.float32:                     # no shrink header: convert on the way out
                    x = x.float()
                return to_nchw(x, C)
            outs = {}
            for name, layer in (("psm", prep["cls"]), ("rm", prep["reg"])):
                y = torch.empty(n_agents, H, W, layer["cout"], device=dev, dtype=torch.float32)
                self._conv(x, layer, n_agents, H, W, y, layer["cout"], 0, False, True, prec, stream)
                outs[name] = to_nchw(y, layer["cout"])
            return outs
