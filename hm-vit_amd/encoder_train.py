"""PointPillar in training mode on libhmvit: ``PointPillar.forward`` under ``nn.Module.train()`` (batch-statistics BatchNorm,
gradients to every parameter), for the un-frozen LiDAR branch of the reference's train loop (``train_camera.py:118-131``
without ``--fix_lidar_backbone``).  Same layer sequence as the inference path (``point_pillar.py:35-54``):

  PillarVFE / PFNLayer   ``pillar_vfe.py:31-53,105-146``   point augmentation (parameter-free indexing arithmetic, torch) ->
                                                           Linear(10 -> 64) [``hmvit_linear`` split; weight gradient ``hmvit_gemm_tn``]
                                                           -> BatchNorm1d on batch statistics + ReLU [``hmvit_bn_train_*``] -> max over
                                                           the 32 points and scatter to the canvas (torch reduction / indexing: their
                                                           adjoints are a routing of the gradient, no arithmetic)
  BaseBEVBackbone        ``base_bev_backbone.py:89-122``   blocks of [3x3 stride-2, 3x3 ...] convolutions and k = s transposed
                                                           convolutions, each followed by BatchNorm2d (eps 1e-3) + ReLU
                                                           [``tail_train.Conv3x3`` / ``Deconv`` / ``BnRelu``], channel concat
  DownsampleConv         ``downsample_conv.py:20-51``      3x3 stride-2 + ReLU, 3x3 + ReLU (bias, no norm)

Every multiply-accumulate over pixels / points runs in libhmvit (f32 maps, split-f16 products); layout changes, zero
insertion, concatenation, max and scatter are torch tensor plumbing on the autograd tape.
"""
from __future__ import annotations

import torch

from . import _lib
from .tail_train import BnRelu, Conv3x3, _stream, bn_relu_module


class Deconv(torch.autograd.Function):
    """``ConvTranspose2d(cin, cout, k = s, stride = s, bias=False)`` on NHWC maps: x (n, H, W, Cin), weight (Cin, Cout, s, s) ->
    (n, s H, s W, Cout).  Forward = the convolution kernel's 1x1-GEMM-with-scatter mode; backward on the space-to-depth view of
    the output gradient: dx = dY_s2d W (``hmvit_linear``), dW = dY_s2d^T x (``hmvit_gemm_tn``)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        ci, co, s, _ = weight.shape
        n, H, W, _ = x.shape
        wn = weight.detach().permute(2, 3, 1, 0).reshape(s * s * co, ci).contiguous()
        y = torch.empty(n, H * s, W * s, co, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.hmvit_conv2d(x.data_ptr(), wn.data_ptr(), None, y.data_ptr(), n, H, W, ci, co, 1, 1, 0, 0, co, 0, s, 1,
                                             _lib.PREC_SPLIT, _stream(x.device)), "conv2d(deconv)")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        ci, co, s, _ = weight.shape
        n, H, W, _ = x.shape
        M, K = n * H * W, s * s * co
        dev = x.device
        dy, un = _lib.grad_pow2(dy)                      # split-f16 products: run on dy 2^k, results times 2^-k (exact)
        dys = dy.reshape(n, H, s, W, s, co).permute(0, 1, 3, 2, 4, 5).reshape(M, K).contiguous()      # [(a, b, co)] per input pixel
        with torch.cuda.device(dev):
            wm = weight.detach().permute(0, 2, 3, 1).reshape(ci, K).contiguous()                         # (ci, (a, b, co))
            dx = torch.empty(n, H, W, ci, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib.hmvit_linear(dys.data_ptr(), wm.data_ptr(), None, None, dx.data_ptr(), M, ci, K, 0, 1, _lib.PREC_SPLIT,
                                             _stream(dev)), "linear(deconv dgrad)")
            dw = torch.zeros(K, ci, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib.hmvit_gemm_tn(dys.data_ptr(), x.data_ptr(), dw.data_ptr(), None, M, K, ci, K, ci, _stream(dev)), "gemm_tn")
        return dx * un, dw.view(s, s, co, ci).permute(3, 2, 0, 1).contiguous() * un


class LinearNoBias(torch.autograd.Function):
    """y = x W^T for a parameter-free input x (M, K) and W (N, K) with any K (zero-padded to the GEMM's 64-wide slabs): the PFN
    layer's Linear(10 -> 64).  Only W receives a gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        M, K = x.shape
        N = weight.shape[0]
        Kp = (K + 63) // 64 * 64
        dev = x.device
        xp = torch.zeros(M, Kp, device=dev, dtype=torch.float32)
        xp[:, :K] = x
        wp = torch.zeros(N, Kp, device=dev, dtype=torch.float32)
        wp[:, :K] = weight.detach()
        y = torch.empty(M, N, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.hmvit_linear(xp.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), M, N, Kp, 0, 1, _lib.PREC_SPLIT,
                                             _stream(dev)), "linear")
        ctx.save_for_backward(xp)
        ctx.K = K
        return y

    @staticmethod
    def backward(ctx, dy):
        (xp,) = ctx.saved_tensors
        dy, un = _lib.grad_pow2(dy)
        M, Kp = xp.shape
        N = dy.shape[1]
        dw = torch.zeros(N, Kp, device=dy.device, dtype=torch.float32)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.lib.hmvit_gemm_tn(dy.data_ptr(), xp.data_ptr(), dw.data_ptr(), None, M, N, Kp, N, Kp, _stream(dy.device)),
                       "gemm_tn")
        return None, dw[:, :ctx.K].contiguous() * un


def pfn_features(vf, vc, vn, voxel_size, lidar_range):
    """pillar_vfe.py:105-141: (Nv, 32, 4) points -> (Nv, 32, 10) [x, y, z, i, offsets from the pillar's point mean, offsets from
    the pillar's centre], padded points zeroed AFTER the augmentation."""
    vx, vy, vz = voxel_size
    x_off, y_off, z_off = vx / 2 + lidar_range[0], vy / 2 + lidar_range[1], vz / 2 + lidar_range[2]
    xyz = vf[:, :, :3]
    mean = xyz.sum(1, keepdim=True) / vn.to(xyz.dtype).view(-1, 1, 1)
    centre = torch.stack([vc[:, 3].to(xyz.dtype) * vx + x_off, vc[:, 2].to(xyz.dtype) * vy + y_off,
                          vc[:, 1].to(xyz.dtype) * vz + z_off], dim=-1)
    feats = torch.cat([vf, xyz - mean, xyz - centre[:, None, :]], dim=-1)
    valid = vn.view(-1, 1).int() > torch.arange(vf.shape[1], dtype=torch.int, device=vf.device).view(1, -1)
    return feats * valid.unsqueeze(-1).to(feats.dtype)


def pointpillar_train_forward(net, data_dict):
    """``PointPillar.forward`` (features only: the HM-ViT model calls it after ``set_return_features()``) in training mode."""
    if not net.return_features:
        raise NotImplementedError("PointPillar in training mode: the feature path (set_return_features()) is built")
    lidar = data_dict["processed_lidar"]
    vf = lidar["voxel_features"].detach().float().contiguous()
    vc = lidar["voxel_coords"].detach().long()
    vn = lidar["voxel_num_points"].detach()
    if vf.device.type != "cuda":
        raise RuntimeError("hm-vit_amd runs on the GPU only (HIP kernels, no CPU fallback)")
    if vf.dim() != 3 or tuple(vf.shape[1:]) != (32, 4) or vc.dim() != 2 or vc.shape[1] != 4 or vn.shape[0] != vf.shape[0]:
        raise ValueError("voxel_features must be (Nv, 32, 4), voxel_coords (Nv, 4) [agent, z, y, x] and voxel_num_points (Nv,)")
    nx, ny, _ = [int(v) for v in net.scatter_cfg["grid_size"]]
    n_agents = int(data_dict["n_agents"]) if "n_agents" in data_dict else int(vc[:, 0].max().item()) + 1
    ok = (vc[:, 0] >= 0) & (vc[:, 0] < n_agents) & (vc[:, 2] >= 0) & (vc[:, 2] < ny) & (vc[:, 3] >= 0) & (vc[:, 3] < nx) & (vc[:, 1] == 0)
    Nv = vf.shape[0]

    # ---- PillarVFE + scatter ----
    pfn = net.pillar_vfe.pfn_layers[0]
    feats = pfn_features(vf, vc, vn, net.args["voxel_size"], net.args["lidar_range"])            # (Nv, 32, 10), no gradient
    h = LinearNoBias.apply(feats.reshape(Nv * 32, 10), pfn.linear.weight)                        # (Nv * 32, 64)
    h = bn_relu_module(h, pfn.norm)                                                              # BatchNorm1d over all points
    pillar = h.view(Nv, 32, 64).max(dim=1)[0]                                                    # (Nv, 64)
    canvas = torch.zeros(n_agents * ny * nx, 64, device=vf.device, dtype=torch.float32)
    flat = (vc[:, 0] * ny + vc[:, 2]) * nx + vc[:, 3]
    canvas = canvas.index_put((flat[ok],), pillar[ok])                                           # NHWC canvas (n, ny, nx, 64)
    x = canvas.view(n_agents, ny, nx, 64)

    # ---- backbone ----
    ups = []
    for blk, de in zip(net.backbone.blocks, net.backbone.deblocks):
        x = bn_relu_module(Conv3x3.apply(x, blk[1].weight, None, blk[1].stride[0]), blk[2])     # ZeroPad2d(1) + conv(pad 0)
        k = 4
        while k < len(blk):
            x = bn_relu_module(Conv3x3.apply(x, blk[k].weight, None, 1), blk[k + 1])
            k += 3
        ups.append(bn_relu_module(Deconv.apply(x, de[0].weight), de[1]))
    x = torch.cat(ups, dim=-1)

    # ---- shrink header ----
    if net.shrink_flag:
        for dc in net.shrink_conv.layers:
            c0, c1 = dc.double_conv[0], dc.double_conv[2]
            if c0.kernel_size != (3, 3) or c0.padding != (1, 1) or c1.kernel_size != (3, 3):
                raise NotImplementedError("shrink header in training mode: 3x3 convolutions with padding 1")
            x = torch.relu(Conv3x3.apply(x, c0.weight, c0.bias, c0.stride[0]))
            x = torch.relu(Conv3x3.apply(x, c1.weight, c1.bias, 1))
    return x.permute(0, 3, 1, 2).contiguous()
