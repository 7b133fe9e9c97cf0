"""ctypes binding of libhmvit.so (include/hmvit.h).  There is no CPU fallback: importing the
package without the built library raises, so a GPU box can never silently run something else."""
from __future__ import annotations

import ctypes as C
import os

# torch first, on purpose: PyTorch-ROCm ships its own libamdhip64.  Loaded first, it is the HIP runtime libhmvit.so's
# DT_NEEDED entry resolves to (same SONAME), so the library and torch share one runtime - streams and device pointers
# pass between them.  Loaded second, the process would hold two runtimes and every call on a torch stream would fail
# with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMVIT_LIB", os.path.join(_HERE, "libhmvit.so"))   # HMVIT_LIB: ablation builds of tools/probe

ABI_VERSION = 12
PREC_F32, PREC_F16, PREC_SPLIT, PREC_MIXED = 0, 1, 2, 3
PART_WINDOW, PART_GRID = 0, 1
NUM_TYPES = 2
MAX_AGENTS = 8
PHASES = ("layout_in", "ln_attn", "qkv_gemm", "attention", "out_proj", "ln_ffn", "ffn1", "ffn2", "head",
          "layout_out")

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)


class StageWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln_gamma", "ln_beta", "w_q", "b_q", "w_kv", "b_kv", "bias_frag", "w_o", "b_o",
        "ffn_ln_gamma", "ffn_ln_beta", "w_1", "b_1", "w_2", "b_2", "img_q", "img_kv", "img_o", "img_ffn", "scales")]


class StageScales(C.Structure):
    """HmvitStageScales (include/hmvit.h): the power-of-two range normalisation of the split-operand modes."""
    _fields_ = [("c_q", C.c_float * 2), ("c_k", (C.c_float * 2) * 2), ("c_v", (C.c_float * 2) * 2), ("k_logit", C.c_float),
                ("c_o", C.c_float * 2), ("c_1", C.c_float * 2), ("s_g", C.c_float * 2), ("k_2", C.c_float * 2)]


class HeadScales(C.Structure):
    _fields_ = [("w1", C.c_float * 2), ("w2", C.c_float * 2), ("l1", C.c_float * 2), ("b1max", C.c_float * 2)]


class FusionDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("L", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("heads", C.c_int32), ("dim_head", C.c_int32), ("window", C.c_int32),
        ("mlp_dim", C.c_int32), ("num_iters", C.c_int32), ("precision", C.c_int32),
        ("apply_head", C.c_int32), ("skip_masked", C.c_int32),
        ("discrete_ratio", C.c_float), ("downsample_rate", C.c_float),
        ("mode", c_i32p), ("record_len", c_i32p), ("cav_mask", c_i32p),
        ("x", C.c_void_p), ("pairwise_t", C.c_void_p), ("out", C.c_void_p),
        ("stage", StageWeights * 2),
        ("head_w1", C.c_void_p), ("head_b1", C.c_void_p), ("head_w2", C.c_void_p),
        ("head_b2", C.c_void_p), ("head_img_ffn", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("parallel", C.c_int32),
        ("split_fc1", C.c_void_p), ("split_ln_g", C.c_void_p), ("split_ln_b", C.c_void_p),
        ("split_fc2", C.c_void_p),
        ("head_scales", C.c_void_p),
        ("self_identity", C.c_int32),
        ("rigid_patch", C.c_int32),
    ]


class StageGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln_gamma", "ln_beta", "w_q", "b_q", "w_kv", "b_kv", "bias_frag", "w_o", "b_o",
        "ffn_ln_gamma", "ffn_ln_beta", "w_1", "b_1", "w_2", "b_2")]


class FusionTrainDesc(C.Structure):
    _fields_ = [("fwd", FusionDesc), ("drop_p", C.c_float), ("seed", C.c_uint64), ("saved", C.c_void_p),
                ("saved_bytes", C.c_size_t), ("bias_frag_neg", C.c_void_p * 2), ("only_stage", C.c_int32),
                ("recompute", C.c_int32)]


class HmvitError(RuntimeError):
    pass


_SIGNATURES = {
    "hmvit_abi_version": (C.c_int, []),
    "hmvit_last_error": (C.c_char_p, []),
    "hmvit_fusion_workspace_bytes": (C.c_size_t, [C.POINTER(FusionDesc)]),
    "hmvit_fusion_forward": (C.c_int, [C.POINTER(FusionDesc), C.c_void_p]),
    "hmvit_fusion_profile": (C.c_int, [C.POINTER(FusionDesc), C.c_void_p, c_f32p, c_i32p]),
    "hmvit_fusion_profile_items": (C.c_int, [c_i32p, c_i32p, C.c_int]),
    "hmvit_fusion_train_saved_bytes": (C.c_size_t, [C.POINTER(FusionTrainDesc)]),
    "hmvit_fusion_backward_workspace_bytes": (C.c_size_t, [C.POINTER(FusionTrainDesc)]),
    "hmvit_fusion_train_forward": (C.c_int, [C.POINTER(FusionTrainDesc), C.c_void_p]),
    "hmvit_fusion_backward": (C.c_int, [C.POINTER(FusionTrainDesc), C.c_void_p, C.c_void_p, C.POINTER(StageGrads)] +
                              [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]),
    "hmvit_dropout_mask": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_float, C.c_void_p]),
    "hmvit_gemm_tn": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "hmvit_linear16": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 2 + [C.c_void_p] * 2),
    "hmvit_bn_train_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_bn_train_stats_centered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_bn_train_apply": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_bn_train_backward": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_pack_small": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hmvit_nchw_to_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_tokens_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, c_i32p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "hmvit_pair_affines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_float, C.c_void_p]),
    "hmvit_warp_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p]),
    "hmvit_window_attention": (C.c_int, [C.c_void_p] * 6 + [c_i32p, c_i32p, c_i32p, C.c_void_p] +
                               [C.c_int] * 12 + [C.c_void_p]),
    "hmvit_pfn_scatter": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, c_f32p, c_f32p, C.c_int,
                                    C.c_void_p]),
    "hmvit_conv2d": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 14 + [C.c_void_p]),
    "hmvit_box_decode": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int] + [C.c_void_p] * 4 +
                         [C.c_int, C.c_void_p]),
    "hmvit_nms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hmvit_nms_rotated": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_float, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "hmvit_quad_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hmvit_voxelize_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "hmvit_voxelize": (C.c_int, [C.c_void_p, C.c_int, c_f32p, c_f32p, C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 5),
    "hmvit_cvt_embed": (C.c_int, [C.c_int] + [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_void_p]),
    "hmvit_bn_relu_tokens": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "hmvit_cross_attention": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p]),
    "hmvit_attention_bias": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]),
    "hmvit_attention_bias_train": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]),
    "hmvit_attention_bias_backward": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 5 + [C.c_void_p]),
    "hmvit_maxpool2d_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p]),
    "hmvit_cross_attention_train": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    "hmvit_cross_attention_backward": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 6 + [C.c_void_p]),
    "hmvit_layernorm_backward": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "hmvit_gelu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hmvit_gelu_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hmvit_conv_range": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "hmvit_absmax": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "hmvit_conv3x3_image_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "hmvit_conv3x3_image": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hmvit_conv_gemm_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hmvit_conv_gemm_image": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hmvit_conv_weight_image": (C.c_int, [C.c_void_p, C.c_int]),
    "hmvit_conv2d_ex": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 12 + [C.c_void_p]),
    "hmvit_conv2d_rowpack": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 10 + [C.c_void_p]),
    "hmvit_maxpool2d": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 8 + [C.c_void_p]),
    "hmvit_debug_tr16": (C.c_int, [C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C hm-vit_amd/csrc`).  hm-vit_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    got = lib.hmvit_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libhmvit.so ABI {got} != binding ABI {ABI_VERSION}: rebuild")
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.hmvit_last_error().decode(errors="replace")
        if rc == -22:
            raise ValueError(f"{what}: {msg}")
        raise HmvitError(f"{what} failed ({rc}): {msg}")


def i32_array(values):
    arr = (C.c_int32 * len(values))(*[int(v) for v in values])
    return arr


# ---------------------------------------------------------------------------------------------------------------------
# range information of the split-mode convolutions (hmvit_conv_range, include/hmvit.h)
# ---------------------------------------------------------------------------------------------------------------------
_RANGE_BLOCKS = {}


def grad_pow2(dy, target_exp: float = 9.0):
    """(dy 2^k, 2^-k): the incoming gradient with its largest magnitude brought to [2^target_exp, 2^(target_exp + 1)) - [2^9, 2^10) by
    default - by a power of two, and the device
    scalar that undoes it.  Every backward function whose products run on split-f16 operands (x = hi + lo) calls this first and
    multiplies its results by the second value: a backward pass is linear in dy, the scaling is exact, and gradients of a real
    loss (1e-4 ... 1e-7 after the focal loss's normalisation) keep their low halves above f16's 2^-24 floor.  No synchronisation."""
    dy = dy.detach().to(torch.float32)
    amax = dy.abs().max()
    k = torch.where(torch.isfinite(amax) & (amax > 0), target_exp - torch.floor(torch.log2(amax.clamp_min(1e-38))), torch.zeros_like(amax))
    k = k.clamp(-100.0, 100.0)
    return (dy * torch.exp2(k)).contiguous(), torch.exp2(-k)


def _range_slot(device):
    """A zeroed device slot (2 x int32; [0] = f32 bit pattern of max |.|).  Slots come from blocks of 64 that stay alive as long
    as a tensor refers to one of theirs; a block is never reused, so a slot is written by exactly one producer chain."""
    import torch
    blk = _RANGE_BLOCKS.get(device)
    if blk is None or blk[1] >= blk[0].shape[0]:
        blk = [torch.zeros(64, 2, dtype=torch.int32, device=device), 0]
        _RANGE_BLOCKS[device] = blk
    i = blk[1]
    blk[1] += 1
    return blk[0][i]


def conv_range(x, w_absmax: float, y, stream, share_out: bool = False):
    """Before an HMVIT_PREC_SPLIT convolution x -> y: hand the library max |x| (the slot the producer of `x` left on the tensor,
    or one measured here, once per tensor), max |w| (host value from weight preparation) and a fresh slot for max |y|, which
    `y` then carries to its consumers.  share_out: several launches write `y` (channel windows of a concatenation)."""
    xs = range_of(x)
    if xs is None:
        xs = _range_slot(x.device)
        check(lib.hmvit_absmax(x.data_ptr(), x.numel(), xs.data_ptr(), stream), "hmvit_absmax")
        set_range(x, xs)
    ys = range_of(y) if share_out else None
    if ys is None:
        ys = _range_slot(y.device)
        set_range(y, ys)
    check(lib.hmvit_conv_range(xs.data_ptr(), float(w_absmax), ys.data_ptr()), "hmvit_conv_range")


def announce_output_range(y, stream=None):
    """Before a producer kernel that reports max |y| itself (hmvit_pfn_scatter): give it a fresh slot, which `y` then carries to
    its consumers (conv_range finds it with range_of)."""
    ys = _range_slot(y.device)
    set_range(y, ys)
    check(lib.hmvit_conv_range(None, 0.0, ys.data_ptr()), "hmvit_conv_range")


def set_range(t, slot):
    """Attach an absmax slot to a tensor, stamped with the tensor's storage address: a bound is only as good as the bytes it was
    measured on.  (The library's own kernels write `t` through raw pointers, which torch's version counter does not see, so the
    stamp is re-taken by whoever writes: conv_range for an output, the stand-alone absmax pass for an input.)"""
    t._hmvit_absmax = (slot, t.data_ptr(), t._version)


def range_of(t):
    """The slot attached to `t`, or None when there is none or the tensor was modified in place by torch / re-pointed since
    (a stale bound would overflow f16 or drop the low halves of a split convolution silently)."""
    r = getattr(t, "_hmvit_absmax", None)
    if r is None:
        return None
    slot, ptr, version = r
    if ptr != t.data_ptr() or version != t._version:
        t._hmvit_absmax = None
        return None
    return slot


def prescale_weights(w):
    """Split-mode convolution weights -> (w * s, -s) with s the power of two that puts max |w| into [2^13, 2^14): the multiply is
    exact, and `-s` is the `w_absmax` argument that tells the kernel the weights are already scaled (hmvit_conv_range)."""
    import math
    wmax = float(w.abs().max())
    if not (wmax > 0.0) or math.isinf(wmax) or math.isnan(wmax):
        return w, 0.0
    e = max(-40, min(40, math.frexp(wmax)[1] - 1))        # floor(log2 wmax), clamped like pow2_scale (csrc/common.hpp)
    s = 2.0 ** (13 - e)
    return w * s, -s


def conv_image(w_rows, ncols: int, cin: int, k: int, stride: int, pad: int, prec: int, wmax: float = 0.0, deconv: bool = False):
    """LDS ring image of a convolution's prepared weight matrix `w_rows` (Ncols, k k Cin) for the LDS-DMA convolution kernels
    (include/hmvit.h): (tensor, kind) - kind 0 for a 3 x 3 / pad 1 layer of stride 1 or 2 (hmvit_conv3x3_image; split and f16), kind 1
    in the GEMM's column order for every other geometry in split mode (hmvit_conv_gemm_image: other strided layers, 1 x 1, transposed) - or
    None where neither kernel applies (exact-f32 mode, split weights that were not pre-scaled, a K that is not a multiple of the
    slab depth).  Built once per weight version by the modules' prepare steps."""
    import torch
    if prec not in (PREC_SPLIT, PREC_F16) or (prec == PREC_SPLIT and not wmax < 0.0):
        return None
    ktot = w_rows.shape[1]
    if k == 3 and stride in (1, 2) and pad == 1 and not deconv and ncols % 8 == 0:      # the ring kernels: stride 1 and stride 2
        nbytes, kind = int(lib.hmvit_conv3x3_image_bytes(ncols, cin, prec)), 0
    elif prec == PREC_SPLIT:
        nbytes, kind = int(lib.hmvit_conv_gemm_image_bytes(ncols, ktot)), 1
    else:
        return None
    if nbytes == 0:
        return None
    img = torch.empty(nbytes, dtype=torch.uint8, device=w_rows.device)
    with torch.cuda.device(w_rows.device):
        stream = C.c_void_p(torch.cuda.current_stream(w_rows.device).cuda_stream)
        if kind == 0:
            check(lib.hmvit_conv3x3_image(w_rows.data_ptr(), ncols, cin, prec, img.data_ptr(), stream), "hmvit_conv3x3_image")
        else:
            check(lib.hmvit_conv_gemm_image(w_rows.data_ptr(), ncols, ktot, img.data_ptr(), stream), "hmvit_conv_gemm_image")
    return img, kind


def conv3_image(w_rows, cout: int, cin: int, k: int, stride: int, pad: int, prec: int, wmax: float = 0.0):
    """The kind-0 image alone (3 x 3 / stride 1 / pad 1), or None."""
    r = conv_image(w_rows, cout, cin, k, stride, pad, prec, wmax) if (k == 3 and stride == 1 and pad == 1) else None
    return r[0] if r is not None and r[1] == 0 else None


def use_conv_image(img, kind: int = 0):
    """Hand a weight image - a conv_image() result, or a bare kind-0 tensor - to the next hmvit_conv2d / _ex call of this thread
    (no-op for None)."""
    if img is None:
        return
    if isinstance(img, tuple):
        img, kind = img
    check(lib.hmvit_conv_weight_image(img.data_ptr(), kind), "hmvit_conv_weight_image")


def inherit_range(dst, src):
    """`dst` holds a subset / copies of the values of `src` (max pooling of a non-negative map, a view): the same bound applies."""
    r = range_of(src)
    if r is not None:
        set_range(dst, r)
