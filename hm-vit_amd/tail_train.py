"""Training-mode layers of the detection tail on libhmvit: what ``HeteroDecoder`` runs under
``nn.Module.train()`` - 3x3 convolution, BatchNorm2d on batch statistics + ReLU, 1x1 head convolution
(``naive_decoder.py:45-54,80-92``, ``hetero_decoder.py:55-69``) - as ``torch.autograd.Function``s whose forward AND backward are
HIP kernels (f32 maps, products on split-f16 operands):

  * ``conv3x3``   forward ``hmvit_conv2d`` (HMVIT_PREC_SPLIT); data gradient = the same kernel on the spatially flipped,
                  in/out-swapped weights; weight gradient = nine ``hmvit_gemm_tn`` products, one per tap, over zero-bordered
                  copies of the input and of the output gradient (with both in the same padded pixel numbering a tap is a pure
                  offset of the flat pixel index, and the zero border of the gradient map cancels every pair that straddles an
                  image edge); bias gradient = the column sums the first tap's product returns;
  * ``bn_relu``   ``hmvit_bn_train_stats`` / ``_apply`` / ``_backward``; mean, variance and the running-statistics update
                  are a few C-length vector operations in torch (momentum and the unbiased running variance as ``nn.BatchNorm2d``);
  * ``conv1x1``   ``hmvit_linear`` (split) forward and data gradient, ``hmvit_gemm_tn`` for the weight / bias gradient.

Maps travel as NHWC (n, H, W, C) float32 CUDA tensors.  Layout changes, zero padding and the weight re-arrangements are torch
tensor plumbing; every multiply-accumulate over pixels runs in libhmvit.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _conv(x, w_rows, bias, cin, cout, k, pad, stride=1):
    n, H, W, _ = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(n, Ho, Wo, cout, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib.hmvit_conv2d(x.data_ptr(), w_rows.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                     n, H, W, cin, cout, k, stride, pad, 0, cout, 0, 0, 1, _lib.PREC_SPLIT, _stream(x.device)),
               "conv2d")
    return y


class Conv3x3(torch.autograd.Function):
    """x (n, H, W, Cin) NHWC, weight (Cout, Cin, 3, 3), bias (Cout) or None -> (n, Ho, Wo, Cout); padding 1, stride 1 or 2
    (``ZeroPad2d(1)`` + stride-2 convolution without padding, base_bev_backbone.py:41-45, is the stride-2 case).  A strided
    convolution's adjoints are the stride-1 ones applied to the output gradient with zeros inserted between its pixels."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride=1):
        x = x.contiguous()
        co, ci = weight.shape[:2]
        w_rows = weight.detach().permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
        with torch.cuda.device(x.device):
            y = _conv(x, w_rows, bias.detach().contiguous() if bias is not None else None, ci, co, 3, 1, stride)
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.has_bias = stride, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy, un = _lib.grad_pow2(dy)                      # split-f16 products: run on dy 2^k, results times 2^-k (exact)
        co, ci = weight.shape[:2]
        n, H, W, _ = x.shape
        dev = x.device
        if ctx.stride != 1:
            s = ctx.stride
            dyz = torch.zeros(n, H, W, co, device=dev, dtype=torch.float32)
            dyz[:, ::s, ::s][:, :dy.shape[1], :dy.shape[2]] = dy
            dy = dyz
        with torch.cuda.device(dev):
            # dx[p][ci] = sum_{tap, co} dy[p - tap][co] w[co][ci][tap]: a 3x3 convolution of dy with flipped, swapped weights
            w_t = weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(ci, 9 * co).contiguous()
            dx = _conv(dy, w_t, None, co, ci, 3, 1)
            # weight gradient over the padded pixel numbering q = (img, y + 1, x + 1)
            Wp = W + 2
            guard = Wp + 1                                               # |tap offset| <= W + 3
            Mp = n * (H + 2) * Wp
            xbuf = torch.zeros(Mp + 2 * guard, ci, device=dev, dtype=torch.float32)
            xbuf[guard:guard + Mp].view(n, H + 2, Wp, ci)[:, 1:H + 1, 1:W + 1] = x
            dyp = torch.zeros(n, H + 2, Wp, co, device=dev, dtype=torch.float32)
            dyp[:, 1:H + 1, 1:W + 1] = dy
            dw = torch.zeros(9, co, ci, device=dev, dtype=torch.float32)
            db = torch.zeros(co, device=dev, dtype=torch.float32)
            st = _stream(dev)
            for ky in range(3):
                for kx in range(3):
                    off = (ky - 1) * Wp + (kx - 1)
                    a = xbuf[guard + off:]
                    tap = ky * 3 + kx
                    _lib.check(_lib.lib.hmvit_gemm_tn(dyp.data_ptr(), a.data_ptr(), dw[tap].data_ptr(),
                                                      db.data_ptr() if tap == 0 else None, Mp, co, ci, co, ci, st), "gemm_tn")
        return (dx * un if ctx.needs_input_grad[0] else None, dw.view(3, 3, co, ci).permute(2, 3, 0, 1).contiguous() * un,
                db * un if ctx.has_bias else None, None)


class BnRelu(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ ReLU unless relu=False) on an NHWC map.  Returns (y, batch mean, biased batch variance); the
    statistics are not differentiable outputs (the caller updates the running buffers with them)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu=True):
        x = x.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dev = x.device
        with torch.cuda.device(dev):
            sums = torch.zeros(2 * C, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib.hmvit_bn_train_stats(x.data_ptr(), sums.data_ptr(), M, C, _stream(dev)), "bn_train_stats")
            mean = sums[:C] / M
            # second pass around the mean: E[x^2] - mean^2 cancels catastrophically for channels whose |mean| dwarfs their spread
            # (behind a convolution bias; padded points of the PFN) and the clamp would hide it
            sums2 = torch.zeros(2 * C, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib.hmvit_bn_train_stats_centered(x.data_ptr(), mean.contiguous().data_ptr(), sums2.data_ptr(), M, C, _stream(dev)),
                       "bn_train_stats_centered")
            r = sums2[:C] / M
            mean = mean + r
            var = (sums2[C:] / M - r * r).clamp_min(0.0)
            rstd = torch.rsqrt(var + eps)
            y = torch.empty_like(x)
            g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
            _lib.check(_lib.lib.hmvit_bn_train_apply(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), b.data_ptr(),
                                                     y.data_ptr(), M, C, 1 if relu else 0, _stream(dev)), "bn_train_apply")
        ctx.save_for_backward(x, y, mean, rstd, g)
        ctx.relu = bool(relu)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, y, mean, rstd, g = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        M = x.numel() // C
        dev = x.device
        with torch.cuda.device(dev):
            sums = torch.zeros(2 * C, device=dev, dtype=torch.float32)
            dx = torch.empty_like(x)
            _lib.check(_lib.lib.hmvit_bn_train_backward(x.data_ptr(), y.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                        g.data_ptr(), sums.data_ptr(), dx.data_ptr(), M, C, 1 if ctx.relu else 0,
                                                        _stream(dev)), "bn_train_backward")
        return dx, sums[C:].clone(), sums[:C].clone(), None, None


class Conv1x1(torch.autograd.Function):
    """x (n, H, W, Cin) NHWC, weight (Cout, Cin, 1, 1), bias (Cout) -> (n, H, W, Cout): the cls / reg heads."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        co, ci = weight.shape[:2]
        M = x.numel() // ci
        w2 = weight.detach().reshape(co, ci).contiguous()
        y = torch.empty(*x.shape[:-1], co, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.hmvit_linear(x.data_ptr(), w2.data_ptr(), bias.detach().contiguous().data_ptr(), None, y.data_ptr(),
                                             M, co, ci, 0, 1, _lib.PREC_SPLIT, _stream(x.device)), "linear")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy, un = _lib.grad_pow2(dy)
        co, ci = weight.shape[:2]
        M = x.numel() // ci
        dev = x.device
        cop = (co + 63) // 64 * 64                       # the split GEMM contracts over a multiple of 64: zero-pad the head width
        with torch.cuda.device(dev):
            dyp = dy.reshape(M, co)
            if cop != co:
                dyp = torch.nn.functional.pad(dyp, (0, cop - co))
            wt = torch.zeros(ci, cop, device=dev, dtype=torch.float32)
            wt[:, :co] = weight.detach().reshape(co, ci).t()
            dx = torch.empty_like(x)
            _lib.check(_lib.lib.hmvit_linear(dyp.contiguous().data_ptr(), wt.data_ptr(), None, None, dx.data_ptr(), M, ci, cop, 0, 1,
                                             _lib.PREC_SPLIT, _stream(dev)), "linear(dgrad)")
            cq = (co + 3) // 4 * 4                       # gemm_tn wants N % 4 == 0
            dy4 = dy.reshape(M, co) if cq == co else torch.nn.functional.pad(dy.reshape(M, co), (0, cq - co))
            dy4 = dy4.contiguous()
            dw = torch.zeros(cq, ci, device=dev, dtype=torch.float32)
            db = torch.zeros(cq, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib.hmvit_gemm_tn(dy4.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), M, cq, ci, cq, ci,
                                              _stream(dev)), "gemm_tn")
        return dx * un, dw[:co].reshape(co, ci, 1, 1).contiguous() * un, db[:co].contiguous() * un


def bn_relu_module(x, bn: torch.nn.BatchNorm2d, relu: bool = True):
    """``bn`` (train mode) + ReLU (unless relu=False) on an NHWC map through ``BnRelu``, with ``nn.BatchNorm2d``'s running-statistics
    update (momentum, unbiased variance, ``num_batches_tracked``)."""
    y, mean, var = BnRelu.apply(x, bn.weight, bn.bias, bn.eps, relu)
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            M = x.numel() // x.shape[-1]
            bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var * (M / max(M - 1, 1)), alpha=m)
    return y
