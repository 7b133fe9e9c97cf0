"""Training through the HIP fusion path: ``torch.autograd.Function`` over ``hmvit_fusion_train_forward`` /
``hmvit_fusion_backward`` (include/hmvit.h), the reference's loss and the step of its train loop.

What the reference does (citations into /root/reference):
  * ``train_camera.py:163-199``   per batch: ``model(batch['ego'])`` -> ``criterion(output, label_dict)`` -> ``backward()``
                                  -> ``optimizer.step()``; DDP all-reduces the gradients (``:126-131``);
  * ``hetero_fusion.py:65-66``, ``base_transformer.py:186-192``: Dropout after the attention out-projection and inside the FFN;
  * ``loss/point_pillar_loss.py:68-142``: focal classification loss + smooth-L1 regression loss with the sine trick;
  * yaml ``optimizer`` block (``opcl/bevformer_point_pillar_hetero.yaml:164-169``): AdamW, lr 2e-4, eps 1e-10, wd 1e-2.

Here the fusion's forward and backward are HIP kernels (csrc/capi_train.hip, csrc/train.hip); the parameter folds
(relation matrices into the K / V projections, the bias table into MFMA fragments, hm-vit_amd/weights.py) stay on the autograd
tape as small tensor algebra, so the kernels' gradients with respect to the folded tensors reach the reference's parameters
(``relation_att``, ``k_linears`` ...) by ordinary autograd.  The loss is plain elementwise torch (kept in
PyTorch), the optimiser is ``torch.optim.AdamW`` as in the reference, the gradient all-reduce is torch's
``DistributedDataParallel`` over RCCL ("nccl") exactly as ``train_camera.py`` wraps the model.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import _lib, weights

STAGE_KEYS = ("ln_gamma", "ln_beta", "w_q", "b_q", "w_kv", "b_kv", "bias_frag", "w_o", "b_o", "ffn_ln_gamma", "ffn_ln_beta",
              "w_1", "b_1", "w_2", "b_2")
HEAD_KEYS = ("head_w1", "head_b1", "head_w2", "head_b2")
N_FOLDED = 2 * len(STAGE_KEYS) + len(HEAD_KEYS)


_TYPED_KEY = re.compile(r"(?:net|_linears)\.(\d)\.")


def folded_for_training(module, block_types=None, head_types=None) -> List[torch.Tensor]:
    """The f32 folded tensors of a HeteroFusion module in the order [stage 0 keys, stage 1 keys, head keys] plus the two
    bias_frag_neg tensors (no gradient), computed from the LIVE parameters (autograd tape kept).

    ``block_types`` / ``head_types``: the agent types whose typed sub-modules the reference would call for this batch - every
    value in ``mode`` (padding slots included: they carry type 0, ``base_transformer.py:138-177``) for the block, the egos'
    types for ``mlp_head`` (``bevformer_point_pillar_hetero.py:47-48``).  Parameters of the other type are folded detached,
    so their ``.grad`` stays ``None`` as it does in the reference (no weight decay on them, "unused" for DDP)."""
    sd = dict(module.named_parameters())
    sd.update(dict(module.named_buffers()))
    head_prefix = module._head_prefix + "."
    for k in list(sd):
        m = _TYPED_KEY.search(k)
        if m is None:
            continue
        live = head_types if k.startswith(head_prefix) else block_types
        if live is not None and int(m.group(1)) not in live:
            sd[k] = sd[k].detach()
    blk = module._block_cfg
    out, neg = [], []
    for which in ("window", "grid"):
        f = weights.fold_stage(sd, module._block_prefix, which, blk["dim_head"], blk["window_size"], torch.float32, keep_graph=True)
        out += [f[k].contiguous() for k in STAGE_KEYS]
        neg.append(f["bias_frag_neg"].contiguous())
    h = weights.fold_head(sd, module._head_prefix, torch.float32, keep_graph=True)
    out += [h[k].contiguous() for k in HEAD_KEYS]
    return out, neg


def recompute_bits(module) -> int:
    """Memory for time in training (HmvitFusionTrainDesc::recompute): ``module.train_recompute`` (or HMVIT_TRAIN_RECOMPUTE in the
    environment) = 1: the FFN pre-activations are not kept for the backward pass, 3: neither are the queries - the backward recomputes
    them with the forward's own kernels (the same rows bit for bit).  cfg2 (5 agents, 200x704, C=256): peak 32.8 -> 30.1 -> 27.4 GiB
    for 92.8 -> 94.4 -> 96.8 ms per step, i.e. +1.6 ms with bit 0 and +4.0 ms with both bits, cumulative (HISTORY.md 12.8, measured
    with tests/tools/train_bench.py on one box).  (``train_recompute_ffn = True`` is the same as 1.)
    The bits are read ONCE per step, here, by the forward: they are part of what the autograd context pins (the saved area's layout
    depends on them), so changing the attribute or the environment between a forward and its backward cannot move the offsets."""
    v = getattr(module, "train_recompute", None)
    if v is None and getattr(module, "train_recompute_ffn", None):
        v = 1
    if v is None:
        v = int(os.environ.get("HMVIT_TRAIN_RECOMPUTE", "0") or 0)
    return int(v) & 3


def _build_desc(module, x, pw, mode_h, rl_h, mask_h, folded, neg, drop_p, seed, out, saved, workspace, only_stage=0, recompute=None):
    blk = module._block_cfg
    B, L, C, H, W = x.shape
    t = _lib.FusionTrainDesc()
    d = t.fwd
    d.B, d.L, d.C, d.H, d.W = B, L, C, H, W
    d.heads, d.dim_head = C // blk["dim_head"], blk["dim_head"]
    d.window, d.mlp_dim, d.num_iters = blk["window_size"], blk["mlp_dim"], module.num_iters
    d.precision, d.apply_head, d.skip_masked = _lib.PREC_F32, (1 if only_stage == 0 else 0), 1
    t.only_stage = only_stage
    t.recompute = recompute_bits(module) if recompute is None else int(recompute)
    d.self_identity = 1            # checked by fusion_forward_with_grad: pairwise_t_matrix[b, i, i] = I
    d.discrete_ratio, d.downsample_rate = float(module.discrete_ratio), float(module.downsample_rate)
    keep = (_lib.i32_array(mode_h), _lib.i32_array(rl_h), _lib.i32_array(mask_h))
    d.mode, d.record_len, d.cav_mask = keep
    d.x, d.pairwise_t = x.data_ptr(), pw.data_ptr()
    d.out = out.data_ptr() if out is not None else None
    n = len(STAGE_KEYS)
    for s in range(2):
        for i, k in enumerate(STAGE_KEYS):
            setattr(d.stage[s], k, folded[s * n + i].data_ptr())
        t.bias_frag_neg[s] = neg[s].data_ptr()
    for i, k in enumerate(HEAD_KEYS):
        setattr(d, k, folded[2 * n + i].data_ptr())
    t.drop_p, t.seed = float(drop_p), int(seed)
    if saved is not None:
        t.saved, t.saved_bytes = saved.data_ptr(), saved.numel()
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel()
    return t, keep


# level of the backward pass: max |d_out| is brought to [2^l, 2^(l + 1)) at the entry (exact; every kernel picks its own operand scales after that)
BACKWARD_LEVEL = 9.0


class FusionTrainFunction(torch.autograd.Function):
    """y = HeteroFusion(x) with gradients for x and for the folded weights.  Inputs after ``ctx_args`` are tensors only.
    ``host`` = (mode, record_len, mask) host lists, optionally followed by ``only_stage`` (1 = the window stage, 2 = the grid stage of
    the block on its own: y is (B, L, C, H, W), every agent an ego, no mlp_head - the branches of the parallel block)."""

    @staticmethod
    def forward(ctx, module, host, drop_p, seed, x, pw, neg0, neg1, *folded):
        only_stage = host[3] if len(host) > 3 else 0
        host = host[:3]
        mode_h, rl_h, mask_h = host
        x = x.detach().to(torch.float32).contiguous()
        pw = pw.detach().to(device=x.device, dtype=torch.float32).contiguous()
        folded = [f.detach().contiguous() for f in folded]
        neg = [neg0.detach().contiguous(), neg1.detach().contiguous()]
        B, L, C, H, W = x.shape
        out = torch.empty((B, C, H, W) if only_stage == 0 else (B, L, C, H, W), device=x.device, dtype=torch.float32)
        rec = recompute_bits(module)       # pinned for this step: the backward's descriptor takes it from ctx, not from the module
        probe, keep0 = _build_desc(module, x, pw, mode_h, rl_h, mask_h, folded, neg, drop_p, seed, out, None, None, only_stage, rec)
        need = _lib.lib.hmvit_fusion_train_saved_bytes(ctypes.byref(probe))
        if need == 0:
            _lib.check(-22, "hmvit_fusion_train_saved_bytes")
        saved = torch.empty(need, dtype=torch.uint8, device=x.device)
        blk = module._block_cfg
        scratch = torch.empty(4 * B * L * H * W * max(C, blk["mlp_dim"]) * (2 if rec else 1), dtype=torch.uint8,
                              device=x.device)
        t, keep = _build_desc(module, x, pw, mode_h, rl_h, mask_h, folded, neg, drop_p, seed, out, saved, scratch, only_stage, rec)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.hmvit_fusion_train_forward(ctypes.byref(t), ctypes.c_void_p(stream)), "hmvit_fusion_train_forward")
        ctx.launch = (module, host + (only_stage, rec), drop_p, seed, x, pw, neg, folded, saved)
        return out

    @staticmethod
    def backward(ctx, d_out):
        module, host, drop_p, seed, x, pw, neg, folded, saved = ctx.launch
        mode_h, rl_h, mask_h, only_stage, rec = host
        d_out = d_out.detach().to(torch.float32).contiguous()
        # The backward kernels form their products on split-f16 operands (x = hi + lo, two f16 halves), and every one of them
        # takes its operands at powers of two it picks itself, from the data it is about to multiply: k_linear16 per token row,
        # k_gemm_tn_split / k_gemm_split per slab (the accumulator follows by exact rescaling), k_attention_bwd per workgroup for dO
        # with a target derived from an a-priori bound on |V'| (the stage's weights, launch_v_bound) so that the one DERIVED operand,
        # dS = P o (dP - D), cannot leave f16's range.  The pass is therefore in range a priori for gradients of any magnitude -
        # including growth by 1e3 ... 1e6 INSIDE the pass behind large FFN weights, which round 4 detected afterwards (non-finite
        # results, one host read per backward) and repaired by re-running the whole pass from a lower level.  One pass, no host
        # synchronisation.  d_out is still brought to max |.| in [2^9, 2^10) first (exact; the pass is linear in d_out): it keeps
        # the f32 intermediates of a 1e-7 loss away from the denormals.
        t, keep = _build_desc(module, x, pw, mode_h, rl_h, mask_h, folded, neg, drop_p, seed, None, saved, None, only_stage, rec)
        need = _lib.lib.hmvit_fusion_backward_workspace_bytes(ctypes.byref(t))
        if need == 0:
            _lib.check(-22, "hmvit_fusion_backward_workspace_bytes")
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        n = len(STAGE_KEYS)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        d_out, unscale = _lib.grad_pow2(d_out, BACKWARD_LEVEL)
        grads = [torch.zeros_like(f) for f in folded]
        d_x = torch.empty_like(x)
        sg = (_lib.StageGrads * 2)()
        for s in range(2):
            for i, k in enumerate(STAGE_KEYS):
                setattr(sg[s], k, grads[s * n + i].data_ptr())
        hg = grads[2 * n:]
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib.hmvit_fusion_backward(ctypes.byref(t), d_out.data_ptr(), d_x.data_ptr(), sg, hg[0].data_ptr(),
                                                      hg[1].data_ptr(), hg[2].data_ptr(), hg[3].data_ptr(), ws.data_ptr(),
                                                      ws.numel(), ctypes.c_void_p(stream)), "hmvit_fusion_backward")
        ctx.launch = None
        d_x.mul_(unscale)
        for g in grads:
            g.mul_(unscale)
        return (None, None, None, None, d_x, None, None, None) + tuple(grads)


def fusion_forward_with_grad(module, x, pairwise_t_matrix, mode, record_len, mask):
    """HeteroFusion.forward on the autograd tape (called by fusion.HeteroFusion when gradients are needed)."""
    if x.device.type != "cuda":
        raise RuntimeError("hm-vit_amd runs on the GPU only (HIP kernels, no CPU fallback)")
    if module._block_cfg["architect_mode"] not in ("sequential", "parallel"):
        raise ValueError(f"{module._block_cfg['architect_mode']} not implemented")        # hetero_fusion.py:472
    if x.dim() != 5:
        raise ValueError("x must be (B, L, C, H, W)")
    blk = module._block_cfg
    if weights.generic_shape(blk["window_size"], blk["dim_head"]) and not getattr(module, "_warned_generic_train", False):
        import warnings
        warnings.warn(f"hmvit_amd: window_size={blk['window_size']} / dim_head={blk['dim_head']} trains through the generic exact-f32 "
                      f"attention kernels (correct, far slower than the tuned kernels for window 4 / 8, dim_head 32)", stacklevel=3)
        module._warned_generic_train = True
    B, L = x.shape[:2]
    pw = pairwise_t_matrix.to(device=x.device, dtype=torch.float32)
    if tuple(pw.shape) != (B, L, L, 4, 4):
        raise ValueError(f"pairwise_t_matrix must be {(B, L, L, 4, 4)}, got {tuple(pw.shape)}")
    eye = torch.eye(4, device=x.device)
    idx = torch.arange(L, device=x.device)
    mode_h, rl_h, mask_h = module._host_small(mode, record_len, mask)
    if len(mode_h) != B * L or len(mask_h) != B * L or len(rl_h) != B:
        raise ValueError("mode / mask must be (B, L) and record_len (B,)")
    if any(not 1 <= int(n) <= L for n in rl_h):
        raise ValueError(f"record_len entries must lie in [1, {L}], got {list(rl_h)}")
    if not bool((pw[:, idx, idx] == eye).all()):
        raise ValueError("training expects pairwise_t_matrix[b, i, i] to be the identity (as the dataset builds it)")
    folded, neg = folded_for_training(module, block_types={int(v) for v in mode_h},
                                      head_types={int(mode_h[b * L]) for b in range(B)})
    drop_p = float(module._block_cfg["drop_out"]) if module.training else 0.0
    new_seed = lambda: int(torch.randint(0, 2 ** 62, (1,)).item()) if drop_p > 0 else 0
    if module._block_cfg["architect_mode"] == "parallel":
        return _parallel_forward_with_grad(module, x, pw, (mode_h, rl_h, mask_h), folded, neg, drop_p, new_seed)
    seed = new_seed()
    module.last_dropout = (drop_p, seed)
    return FusionTrainFunction.apply(module, (mode_h, rl_h, mask_h), drop_p, seed, x, pw, neg[0], neg[1], *folded)


def _parallel_forward_with_grad(module, x, pw, host, folded, neg, drop_p, new_seed):
    """architect_mode 'parallel' on the tape (hetero_fusion.py:459-470, bevformer_point_pillar_hetero.py:39-49): per iteration the
    window and the grid stage both start from the block input - each ONE call of the training kernels in their single-stage form
    (HmvitFusionTrainDesc::only_stage) - and are merged by SplitAttn (fusion_modules/split_attn.py:32-67): global average pool,
    two small Linears around a LayerNorm + ReLU, a two-way softmax per channel, a weighted sum (Linears / LayerNorm on libhmvit
    through camera_train's Functions, the rest elementwise torch); then mlp_head (typed Linear - GELU - Linear) on the ego row."""
    from .camera_train import GeluFn, LayerNormFn, LinearFn
    mode_h = host[0]
    B, L, C, H, W = x.shape
    sa = module.hetero_fusion_block.split_attn
    seeds = []
    cur = x.to(torch.float32)
    for _ in range(module.num_iters):
        branches = []
        for which in (1, 2):
            seed = new_seed()
            seeds.append(seed)
            branches.append(FusionTrainFunction.apply(module, host + (which,), drop_p, seed, cur, pw, neg[0], neg[1], *folded))
        a, b = branches
        gap = (a + b).mean((3, 4)).reshape(B * L, C)
        g = torch.relu(LayerNormFn.apply(LinearFn.apply(gap, sa.fc1.weight, None), sa.bn1.weight, sa.bn1.bias, sa.bn1.eps))
        w = torch.softmax(LinearFn.apply(g, sa.fc2.weight, None).reshape(B, L, 2, C), dim=2)
        cur = a * w[:, :, 0, :, None, None] + b * w[:, :, 1, :, None, None]
    module.last_dropout = (drop_p, seeds)
    outs = []
    for bi in range(B):
        net = module.mlp_head.net[int(mode_h[bi * L])]
        tok = cur[bi, 0].reshape(C, H * W).t()
        y = LinearFn.apply(GeluFn.apply(LinearFn.apply(tok, net[0].weight, net[0].bias)), net[3].weight, net[3].bias)
        outs.append(y.t().reshape(C, H, W))
    return torch.stack(outs)


# ---------------------------------------------------------------------------------------------
# loss (loss/point_pillar_loss.py:68-142), elementwise torch
# ---------------------------------------------------------------------------------------------
class PointPillarLoss(torch.nn.Module):
    """Focal classification loss (alpha 0.25, gamma 2, normalised by the number of positive anchors) + smooth-L1 regression
    loss (beta 1/9) on the 7 box deltas with sin(a - b) = sin a cos b - cos a sin b on the yaw (``add_sin_difference``),
    ``cls_weight`` 1, ``reg`` coefficient from the yaml (2.0)."""

    def __init__(self, args: dict):
        super().__init__()
        self.alpha, self.gamma = 0.25, 2.0
        self.beta = 1.0 / 9.0
        self.cls_weight = args.get("cls_weight", 1.0)
        self.reg_coe = args.get("reg", 2.0)
        self.loss_dict = {}

    @staticmethod
    def add_sin_difference(boxes1, boxes2, dim=6):
        rad_pred = torch.sin(boxes1[..., dim:dim + 1]) * torch.cos(boxes2[..., dim:dim + 1])
        rad_tg = torch.cos(boxes1[..., dim:dim + 1]) * torch.sin(boxes2[..., dim:dim + 1])
        boxes1 = torch.cat([boxes1[..., :dim], rad_pred, boxes1[..., dim + 1:]], dim=-1)
        boxes2 = torch.cat([boxes2[..., :dim], rad_tg, boxes2[..., dim + 1:]], dim=-1)
        return boxes1, boxes2

    def forward(self, output_dict: Dict[str, torch.Tensor], target_dict: Dict[str, torch.Tensor]):
        rm, psm = output_dict["rm"], output_dict["psm"]
        targets = target_dict["targets"]
        B = psm.shape[0]
        cls_preds = psm.permute(0, 2, 3, 1).contiguous()                     # (B, H, W, A)
        box_cls_labels = target_dict["pos_equal_one"].view(B, -1).contiguous()
        positives = box_cls_labels > 0
        negatives = box_cls_labels == 0
        negative_cls_weights = negatives * 1.0
        cls_weights = (negative_cls_weights + 1.0 * positives).float()
        reg_weights = positives.float()
        pos_normalizer = positives.sum(1, keepdim=True).float()
        reg_weights = reg_weights / torch.clamp(pos_normalizer, min=1.0)
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        cls_targets = box_cls_labels.unsqueeze(-1).float()                   # one-hot over the single class
        cls_preds = cls_preds.view(B, -1, 1)
        # sigmoid focal loss with logits
        pred_sigmoid = torch.sigmoid(cls_preds)
        alpha_weight = cls_targets * self.alpha + (1 - cls_targets) * (1 - self.alpha)
        pt = cls_targets * (1.0 - pred_sigmoid) + (1.0 - cls_targets) * pred_sigmoid
        focal_weight = alpha_weight * torch.pow(pt, self.gamma)
        bce = torch.clamp(cls_preds, min=0) - cls_preds * cls_targets + torch.log1p(torch.exp(-torch.abs(cls_preds)))
        cls_loss = (focal_weight * bce).squeeze(-1) * cls_weights
        conf_loss = cls_loss.sum() / B * self.cls_weight

        rm = rm.permute(0, 2, 3, 1).contiguous().view(B, -1, 7)
        targets = targets.view(B, -1, 7)
        box_preds_sin, reg_targets_sin = self.add_sin_difference(rm, targets)
        reg_targets_sin = torch.where(torch.isnan(reg_targets_sin), box_preds_sin, reg_targets_sin)   # ignore nan targets
        diff = box_preds_sin - reg_targets_sin
        n = torch.abs(diff)
        loc = torch.where(n < self.beta, 0.5 * n ** 2 / self.beta, n - 0.5 * self.beta)
        reg_loss = (loc * reg_weights.unsqueeze(-1)).sum() / B * self.reg_coe
        total = reg_loss + conf_loss
        self.loss_dict = {"total_loss": total, "reg_loss": reg_loss, "conf_loss": conf_loss}
        return total


def make_optimizer(params, cfg: dict | None = None):
    """yaml ``optimizer`` block of the shipped config: AdamW(lr 2e-4, eps 1e-10, weight_decay 1e-2)
    (opcl/bevformer_point_pillar_hetero.yaml:164-169, tools/train_utils.py:206-230)."""
    cfg = cfg or {}
    args = dict(eps=1e-10, weight_decay=1e-2)
    args.update(cfg.get("args", {}))
    return torch.optim.AdamW([p for p in params if p.requires_grad], lr=cfg.get("lr", 2e-4), **args)


def train_step(model, criterion, optimizer, batch, label_dict):
    """One iteration of train_camera.py:163-199: zero_grad -> forward -> loss -> backward (DDP all-reduces here) -> step."""
    model.train()
    optimizer.zero_grad()
    out = model(batch)
    loss = criterion(out, label_dict)
    loss.backward()
    optimizer.step()
    return loss.detach()
