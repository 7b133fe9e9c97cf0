"""GPU tests of the training path (hm-vit_amd/train.py -> hmvit_fusion_train_forward / hmvit_fusion_backward): the HIP
backward pass against torch.autograd through the CPU oracle (plain torch, pinned to the reference by the goldens) on the
same seeded inputs.  Tolerances: forward 1e-4 (exact-f32 mode), gradients 1e-3 rel-max per tensor."""
import ctypes

import pytest
import torch

from conftest import rel_max_err
from oracle import hmvit_oracle as O

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-3


def _net(cfg, sd):
    import hmvit_amd
    net = hmvit_amd.HeteroFusion(cfg, precision="f32")
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def _oracle_grads(cfg, sd, scene, gy, drop_masks=None, dtype=torch.float32, device=None):
    """Output, d/dx and d/dparam of <oracle(x), gy> by torch.autograd through the oracle (dtype=torch.float64: the yardstick
    run).  device="cuda": the oracle's token arithmetic runs through torch on the GPU (its sampling geometry and visibility
    masks stay on the CPU, oracle/hmvit_oracle.py `hetero_fusion(device=)`) - used for the float64 yardstick at sizes where
    the CPU takes minutes; the leaves, and therefore the returned gradients, live on the CPU either way."""
    x, pw, mode, rl, mask = scene
    sd = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    x = x.to(dtype).clone().requires_grad_(True)
    y = O.hetero_fusion(x, pw, mode, rl, mask, sd, cfg, drop_masks=drop_masks, dtype=dtype, device=device)
    (y * gy.to(device=y.device, dtype=dtype)).sum().backward()
    return y.detach().cpu(), x.grad, {k: v.grad for k, v in sd.items() if v.is_floating_point()}


def _check_grads(net, ref_grads, x_grad, ref_x_grad, used_only=True, tol=None):
    """tol: None = GRAD_TOL for every tensor, or {name: bound} (tensors missing from it: GRAD_TOL)."""
    assert rel_max_err(x_grad.cpu(), ref_x_grad) < GRAD_TOL
    worst = {}
    # A gradient that is zero in exact arithmetic comes out as round-off noise on both sides (the key bias when every source
    # has the same type: it shifts all logits of a row alike and the softmax does not see it).  Such tensors are held to an
    # absolute bound instead: 1e-4 of the largest parameter gradient of the model.
    gmax = max(float(g.abs().max()) for g in ref_grads.values() if g is not None)
    for name, p in net.named_parameters():
        ref = ref_grads.get(name)
        if ref is None or float(ref.abs().max()) == 0.0:
            # not reached by the loss in the oracle either (aggregate_fc; the type that does not occur in the scene)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        scale = max(float(ref.abs().max()), 1e-4 * gmax)
        worst[name] = float((p.grad.cpu().double() - ref.double()).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if not v < (GRAD_TOL if tol is None else tol.get(k, GRAD_TOL))}
    assert not bad, bad
    return worst


CASES = [
    # C, window, L, H, W, modes, n_valid, B
    (64, 4, 3, 16, 24, [0, 1, 0], 2, 1),       # g3-sized: one padded agent, mixed types
    (64, 4, 3, 8, 16, [1, 0, 0], 3, 2),        # two samples
    (256, 8, 3, 16, 16, [1, 0, 1], 3, 1),      # g4-sized channel count, window 8
    (128, 8, 2, 16, 24, [1, 1], 2, 1),         # single agent type: the other type's parameters get no gradient
]


@pytest.mark.parametrize("C,w,L,H,W,modes,n_valid,B", CASES)
def test_backward_matches_oracle_autograd(C, w, L, H, W, modes, n_valid, B):
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=4)
    sd = O.random_state_dict(cfg, seed=5)
    scene = O.synthetic_scene(L, C, H, W, modes, n_valid=n_valid, seed=6, B=B, tx_step=3.0, ty_step=-2.0)
    gy = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(7))
    y_ref, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy)

    net = _net(cfg, sd).eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    assert y.requires_grad
    assert rel_max_err(y.detach().cpu(), y_ref) < 1e-4
    (y * gy.cuda()).sum().backward()
    _check_grads(net, gp_ref, x.grad, gx_ref)
    # constructed-but-unused parameters (hetero_fusion.py:326-327) stay without gradient: DDP needs find_unused_parameters
    assert all(p.grad is None for n, p in net.named_parameters() if "aggregate_fc" in n)


def _mask(n, seed, salt, p):
    from hmvit_amd import _lib
    m = torch.empty(n, device="cuda")
    _lib.check(_lib.lib.hmvit_dropout_mask(m.data_ptr(), n, ctypes.c_uint64(seed), salt, p,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dropout_mask")
    return m.cpu()


def test_training_mode_dropout_replayed_through_the_oracle():
    """Train mode: Dropout 0.1 after the out-projection and inside the FFN (hetero_fusion.py:65-66, base_transformer.py:186-192).
    The kernels' masks are a pure function of (seed, slot, stage, element); replaying them through the oracle must reproduce
    the output and all gradients."""
    C, w, L, H, W, B = 64, 4, 3, 16, 16, 1
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=4)
    p = cfg["hetero_fusion_block"]["drop_out"]
    assert p > 0
    sd = O.random_state_dict(cfg, seed=15)
    scene = O.synthetic_scene(L, C, H, W, [1, 0, 1], seed=16, B=B, tx_step=3.0, ty_step=-2.0)
    gy = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(17))
    net = _net(cfg, sd).train()
    x = scene[0].cuda().requires_grad_(True)
    torch.manual_seed(123)
    y = net(x, *[t.cuda() for t in scene[1:]])
    (y * gy.cuda()).sum().backward()
    drop_p, seed = net.last_dropout
    assert drop_p == p
    masks = []
    for it in range(cfg["num_iters"]):
        per_stage = []
        for s in range(2):
            k = 2 * it + s
            triple = []
            for which in range(3):
                m = torch.stack([_mask(H * W * C, seed + 0x51ED270B1 * (slot + 1), 4 * k + which, p).reshape(H, W, C)
                                 for slot in range(B * L)]).reshape(B, L, H, W, C)
                triple.append(m)
            per_stage.append(triple)
        masks.append(per_stage)
    keep_rate = float((masks[0][0][0] != 0).float().mean())
    assert abs(keep_rate - (1 - p)) < 0.02
    assert abs(float(masks[0][0][0].max()) - 1 / (1 - p)) < 1e-6
    y_ref, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy, drop_masks=masks)
    assert rel_max_err(y.detach().cpu(), y_ref) < 1e-4
    _check_grads(net, gp_ref, x.grad, gx_ref)
    # a second step draws a different mask
    y2 = net(x, *[t.cuda() for t in scene[1:]])
    assert not torch.equal(y2, y)


def test_train_mode_without_grad_raises_and_eval_inference_is_untouched():
    cfg = O.make_config(64, 4, 2)
    sd = O.random_state_dict(cfg, seed=3)
    scene = [t.cuda() for t in O.synthetic_scene(2, 64, 8, 8, [1, 1], seed=4)]
    net = _net(cfg, sd).train()
    with torch.no_grad(), pytest.raises(RuntimeError):
        net(*scene)
    net.eval()
    y = net(*scene)                       # parameters require grad, the input does not: inference path
    assert not y.requires_grad
    blk = net.hetero_fusion_block.train()
    with pytest.raises(RuntimeError):
        blk(*scene)


def test_adamw_steps_reduce_the_loss():
    """A few iterations of the reference's step (train_camera.py:163-199 with the yaml's AdamW) on the fusion alone."""
    from hmvit_amd.train import make_optimizer
    cfg = O.make_config(64, 4, 3, voxel=0.4, downsample=4)
    sd = O.random_state_dict(cfg, seed=21)
    scene = [t.cuda() for t in O.synthetic_scene(3, 64, 16, 16, [1, 0, 1], seed=22, tx_step=3.0, ty_step=-2.0)]
    # a reachable target: the output of the same architecture with other weights (teacher / student)
    with torch.no_grad():
        target = _net(cfg, O.random_state_dict(cfg, seed=24)).eval()(*scene)
    net = _net(cfg, sd).eval()
    net.force_autograd = True            # eval mode: no dropout noise in the loss curve, still on the tape
    opt = make_optimizer(net.parameters(), {"lr": 1e-3})
    losses = []
    for _ in range(15):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(*scene), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses
    assert sum(b < a for a, b in zip(losses, losses[1:])) >= 12, losses


def test_ddp_wraps_the_module_over_rccl():
    """train_camera.py:126-131: DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True) -- the unused
    aggregate_fc parameters must not stall the reducer.  One rank here (the box has one GPU): the hooks, the bucket
    all-reduce over RCCL and the custom autograd Function still run end to end."""
    import os
    import socket
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = O.make_config(64, 4, 2, voxel=0.4, downsample=4)
        sd = O.random_state_dict(cfg, seed=31)
        scene = [t.cuda() for t in O.synthetic_scene(2, 64, 8, 16, [1, 0], seed=32, tx_step=2.0, ty_step=1.0)]
        net = _net(cfg, sd).train()
        ref = _net(cfg, sd).train()
        ddp = DistributedDataParallel(net, device_ids=[0], find_unused_parameters=True)
        for _ in range(2):                                   # two iterations: the reducer re-arms
            torch.manual_seed(5)
            ddp(*scene).square().mean().backward()
        torch.manual_seed(5)
        ref(*scene).square().mean().backward()
        torch.manual_seed(5)
        ref(*scene).square().mean().backward()
        for (n, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
            if b.grad is None or float(b.grad.abs().max()) == 0.0:      # unused (aggregate_fc) / the absent agent type
                assert a.grad is None or float(a.grad.abs().max()) == 0.0, n
            else:
                assert rel_max_err(a.grad, b.grad) < 1e-4, n     # f32 atomics: summation order differs run to run
    finally:
        dist.destroy_process_group()


def test_backward_matches_oracle_autograd_five_agents_64x176():
    """BASELINE configs[4]'s fusion at a size where every scheduling path of the training kernels is live:
    5 agents 10110 (camera and LiDAR types, camera-type collaborators), C = 256, window 8, 64 x 176 px = 176 windows per agent,
    against torch.autograd through the oracle in float64 (token arithmetic on the GPU through torch, geometry on the CPU: on the
    CPU alone this took 130 s of the suite's budget)."""
    C, w, L, H, W = 256, 8, 5, 64, 176
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=2)
    sd = O.random_state_dict(cfg, seed=45)
    scene = O.synthetic_scene(L, C, H, W, [1, 0, 1, 1, 0], seed=46, tx_step=6.0, ty_step=-4.0)
    gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(47))
    # The yardstick is the oracle's autograd in FLOAT64; the same in float32 tells how far fp32 arithmetic itself is from it at
    # this size (sums over 56 k tokens): a parameter gradient is held to max(1e-3, 3 x that distance).
    y_ref, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy, dtype=torch.float64, device="cuda")
    _, gx32, gp32 = _oracle_grads(cfg, sd, scene, gy, device="cuda")
    gmax = max(float(g.abs().max()) for g in gp_ref.values() if g is not None)
    noise = {k: float((gp32[k].double() - g).abs().max()) / max(float(g.abs().max()), 1e-4 * gmax)
             for k, g in gp_ref.items() if g is not None and float(g.abs().max()) > 0}
    net = _net(cfg, sd).eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    assert rel_max_err(y.detach().cpu(), y_ref) < 1e-4
    (y * gy.cuda()).sum().backward()
    try:
        worst = _check_grads(net, gp_ref, x.grad, gx_ref, tol={k: max(GRAD_TOL, 3 * v) for k, v in noise.items()})
    finally:
        w = {n: float((p.grad.cpu().double() - gp_ref[n]).abs().max()) / max(float(gp_ref[n].abs().max()), 1e-4 * gmax)
             for n, p in net.named_parameters() if p.grad is not None and gp_ref.get(n) is not None and float(gp_ref[n].abs().max()) > 0}
        top = sorted(w.items(), key=lambda kv: -kv[1])[:6]
        print("\n64x176 / 5 agents, worst parameter gradients vs float64 autograd  [HIP | fp32 oracle]:")
        for k, v in top:
            print(f"   {k}: {v:.2e} | {noise[k]:.2e}")
        print(f"   d/dx: {rel_max_err(x.grad.cpu(), gx_ref):.2e} | {rel_max_err(gx32, gx_ref):.2e}")


@pytest.mark.parametrize("gscale", [1e-4, 1e-6, 1e4])
def test_backward_is_independent_of_the_scale_of_the_loss(gscale):
    """the backward products run on split-f16 operands, and real loss gradients (focal loss normalised by the
    positives) arrive at 1e-4 ... 1e-7, far below f16's comfortable range.  The upstream gradient is renormalised by a power
    of two inside FusionTrainFunction.backward (exact: the pass is linear in it), so the parity must not depend on its scale."""
    C, w, L, H, W = 64, 4, 3, 16, 24
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=4)
    sd = O.random_state_dict(cfg, seed=5)
    scene = O.synthetic_scene(L, C, H, W, [0, 1, 0], n_valid=3, seed=6, tx_step=3.0, ty_step=-2.0)
    gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(7)) * gscale
    _, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy.double().float())
    net = _net(cfg, sd).eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    (y * gy.cuda()).sum().backward()
    _check_grads(net, gp_ref, x.grad, gx_ref)


def test_cfg2_gradient_is_consistent_with_finite_differences():
    """The train step at the headline size (5 x 200 x 704 x 256, where the persistent split attention with its saved
    log-sum-exp, k_warp_adjoint and the batched column sums run at full occupancy) cannot be checked against the CPU oracle's
    autograd in reasonable time, so it is checked against itself: the directional derivative <grad L, d> along three random
    parameter directions d (and one input direction) against the central difference (L(theta + eps d) - L(theta - eps d)) / 2 eps
    of the exact-f32 inference forward, L = <y, g> accumulated in float64.  1e-2."""
    import hmvit_amd
    from hmvit_amd import synthetic as S
    c = dict(L=5, C=256, H=200, W=704, window=8, modes=[1, 0, 1, 1, 0])
    cfg = S.make_config(c["C"], c["window"], c["L"], voxel=0.4, downsample=1)
    net = S.seeded_fusion(cfg, precision="f32", seed=0).cuda().eval()
    scene = [t.cuda() for t in S.synthetic_scene(c["L"], c["C"], c["H"], c["W"], c["modes"], seed=3)]
    g = torch.randn(1, c["C"], c["H"], c["W"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))

    def loss_value():
        with torch.no_grad():
            return float((net(*scene).double() * g.double()).sum())

    net.force_autograd = True
    x = scene[0].clone().requires_grad_(True)
    y = net(x, *scene[1:])
    (y * g).sum().backward()
    net.force_autograd = False
    base = float((y.detach().double() * g.double()).sum())
    # the training forward and the inference forward are the same function (to f32 round-off)
    assert abs(loss_value() - base) <= 1e-5 * float((y.detach().abs().double() * g.abs().double()).sum())
    params = [(n, p) for n, p in net.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0]
    assert len(params) > 40
    gen = torch.Generator(device="cuda").manual_seed(11)
    for trial in range(3):
        dirs = {n: torch.randn(p.shape, device="cuda", generator=gen) * p.detach().abs().mean().clamp_min(1e-3) for n, p in params}
        analytic = sum(float((p.grad.double() * dirs[n].double()).sum()) for n, p in params)
        eps = 2e-3
        with torch.no_grad():
            for n, p in params:
                p.add_(eps * dirs[n])
            up = loss_value()
            for n, p in params:
                p.sub_(2 * eps * dirs[n])
            down = loss_value()
            for n, p in params:
                p.add_(eps * dirs[n])
        fd = (up - down) / (2 * eps)
        print(f"\ncfg2 directional derivative {trial}: analytic {analytic:.6e}  central difference {fd:.6e}")
        assert abs(fd - analytic) <= 1e-2 * max(abs(analytic), abs(fd)), (trial, analytic, fd)
    # one direction in the input
    d = torch.randn(scene[0].shape, device="cuda", generator=gen)
    analytic = float((x.grad.double() * d.double()).sum())
    eps = 1e-2
    with torch.no_grad():
        x0 = scene[0]
        scene[0] = x0 + eps * d
        up = loss_value()
        scene[0] = x0 - eps * d
        down = loss_value()
        scene[0] = x0
    fd = (up - down) / (2 * eps)
    print(f"\ncfg2 directional derivative (input): analytic {analytic:.6e}  central difference {fd:.6e}")
    assert abs(fd - analytic) <= 1e-2 * max(abs(analytic), abs(fd))


@pytest.mark.parametrize("C,w,L,H,W,modes,n_valid", [(64, 4, 3, 16, 24, [0, 1, 0], 2), (256, 8, 3, 16, 24, [1, 0, 1], 3)])
def test_parallel_block_backward_matches_oracle_autograd(C, w, L, H, W, modes, n_valid):
    """architect_mode 'parallel' (hetero_fusion.py:459-470): both stages from the block input, merged by SplitAttn - forward and every
    gradient against torch.autograd through the oracle (VERDICT r2 missing #3)."""
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=4, arch="parallel")
    sd = O.random_state_dict(cfg, seed=35)
    scene = O.synthetic_scene(L, C, H, W, modes, n_valid=n_valid, seed=36, tx_step=3.0, ty_step=-2.0)
    gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(37))
    y_ref, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy)
    net = _net(cfg, sd).eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    assert y.requires_grad and rel_max_err(y.detach().cpu(), y_ref) < 1e-4
    (y * gy.cuda()).sum().backward()
    # the padded agent's input gradient is not comparable (its maps are zeroed between the iterations here, the reference runs
    # them through the FFN; neither reaches the output): compare the real agents
    gx = x.grad.cpu().clone()
    gx[:, n_valid:] = gx_ref[:, n_valid:]
    worst = _check_grads(net, gp_ref, gx, gx_ref)
    assert any("split_attn" in k for k in worst)
    # train mode draws a dropout stream per stage call
    net.train()
    net(x, *[t.cuda() for t in scene[1:]])
    assert len(net.last_dropout[1]) == 2 * cfg["num_iters"]


def _grad_spread(net, scene, gy, runs):
    """Gradients of `runs` identical forward + backward passes: largest relative difference to the first pass, per tensor."""
    def once():
        net.zero_grad()
        x = scene[0].clone().requires_grad_(True)
        y = net(x, *scene[1:])
        (y * gy).sum().backward()
        torch.cuda.synchronize()
        out = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        out["d_x"] = x.grad.detach().clone()
        return out
    first = once()
    worst = {k: 0.0 for k in first}
    for _ in range(runs - 1):
        r = once()
        assert r.keys() == first.keys()
        for k in r:
            worst[k] = max(worst[k], float((r[k] - first[k]).abs().max() / first[k].abs().max().clamp_min(1e-30)))
    return worst


@pytest.mark.parametrize("arch,L,H,W,downsample,tx", [("sequential", 5, 64, 176, 4, 9.0),      # many keys outside a source's view
                                                      ("sequential", 5, 200, 704, 1, 10.0),    # the headline size (cfg2 geometry)
                                                      ("parallel", 5, 64, 176, 4, 9.0)])       # the only_stage form of the kernels
def test_backward_is_reproducible_run_to_run(arch, L, H, W, downsample, tx):
    """ADVICE r3 / DESIGN 10.4: `k_attention_bwd` at two workgroups per CU once produced gradients that differed by 1e-2 in one
    pass out of two when a staging wave held visible and invisible keys side by side.  Eight passes on identical inputs with
    masked keys present: every gradient (d/dx and all parameters) must agree with the first pass to 1e-5 of its largest
    element - the only run-to-run freedom left is the order of a few float atomics in the bias column sums (~1e-6)."""
    C, w = 256, 8
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=downsample, arch=arch)
    sd = O.random_state_dict(cfg, seed=7)
    scene = [t.cuda() for t in O.synthetic_scene(L, C, H, W, [1, 0, 1, 1, 0], n_valid=L, seed=3, tx_step=tx, ty_step=-5.0 * tx / 9.0)]
    gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(5)).cuda()
    net = _net(cfg, sd).eval()
    net.force_autograd = True
    worst = _grad_spread(net, scene, gy, runs=8 if H * W < 100000 else 4)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"\nbackward run-to-run spread [{arch} {L}x{H}x{W}]: " + ", ".join(f"{k.replace('hetero_fusion_block.', '')} {v:.1e}" for k, v in top))
    assert max(worst.values()) <= 1e-5, top


@pytest.mark.parametrize("C,w", [(64, 4), (256, 8)])
@pytest.mark.parametrize("case,xscale,wscale", [("x1e3", 1e3, 1.0), ("x1e-3", 1e-3, 1.0), ("w30", 1.0, 30.0), ("w1e-2", 1.0, 1e-2),
                                                ("w1e-3", 1.0, 1e-3)])
def test_backward_under_input_and_weight_scaling(case, xscale, wscale, C, w):
    """the training kernels form their products on split-f16 operands (x = hi + lo), and f16 has five exponent bits.  Inputs
    and Linear weights far from unit scale (as tests/test_hip_range.py does for inference): forward and every gradient against the
    oracle's autograd in FLOAT64; a tensor is held to max(1e-3, 4 x the fp32 oracle's own distance from float64).  Both Linear
    paths: C = 64 (the generic split GEMM) and C = 256 (the x16 kernels with per-token scaling).  Weights x 30 make the gradients
    grow by 1e3 - 1e6 inside the pass: every backward kernel scales its own operands (per token row / per slab / per workgroup with an
    a-priori bound on |V'|), so the pass is in range a priori - round 4 detected the overflow afterwards and re-ran the pass; that
    retry loop is gone and the module must not carry a retry counter any more."""
    L, H, W = 3, 16, 24
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=4)
    sd = O.random_state_dict(cfg, seed=5)
    for k in sd:
        if ("linears" in k or ".fn.net." in k or k.startswith("mlp_head")) and sd[k].is_floating_point():
            sd[k] = sd[k] * wscale
    scene = list(O.synthetic_scene(L, C, H, W, [0, 1, 0], n_valid=3, seed=6, tx_step=3.0, ty_step=-2.0))
    scene[0] = scene[0] * xscale
    gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(7))
    y_ref, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy, dtype=torch.float64, device="cuda")
    y32, gx32, gp32 = _oracle_grads(cfg, sd, scene, gy)        # the reference arithmetic's own noise: fp32 on the CPU
    gmax = max(float(g.abs().max()) for g in gp_ref.values() if g is not None)
    noise = {k: float((gp32[k].double() - g).abs().max()) / max(float(g.abs().max()), 1e-4 * gmax)
             for k, g in gp_ref.items() if g is not None and float(g.abs().max()) > 0}
    net = _net(cfg, sd).eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    assert torch.isfinite(y).all()
    fwd_noise = rel_max_err(y32, y_ref)
    assert rel_max_err(y.detach().cpu(), y_ref) < max(1e-4, 4 * fwd_noise)
    (y * gy.cuda()).sum().backward()
    assert torch.isfinite(x.grad).all()
    gx_noise = rel_max_err(gx32, gx_ref)
    # weights x 30: the logits reach 1e6 and the problem is ill-conditioned for ANY 24-bit arithmetic (the fp32 oracle itself is
    # 2e-4 ... 7e-4 from float64 there).  A split operand carries 22 bits (hi + lo of 11 each, lo x lo dropped), four times
    # coarser than fp32's 24: measured 2.5 ... 4.2 x the fp32 oracle's own distance - held to 8 x, as before
    k_noise = 8 if wscale > 1 else 4
    assert rel_max_err(x.grad.cpu(), gx_ref) < max(GRAD_TOL, k_noise * gx_noise), case
    worst = {}
    for name, p in net.named_parameters():
        ref = gp_ref.get(name)
        if ref is None or float(ref.abs().max()) == 0.0:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        worst[name] = float((p.grad.cpu().double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-4 * gmax)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"\ntrain range[{case}]: forward {rel_max_err(y.detach().cpu(), y_ref):.1e} (fp32 oracle {fwd_noise:.1e}), d/dx "
          f"{rel_max_err(x.grad.cpu(), gx_ref):.1e} ({gx_noise:.1e}), worst parameter gradients",
          [(k.replace("hetero_fusion_block.", ""), f"{v:.1e}", f"{noise.get(k, 0):.1e}") for k, v in top])
    bad = {k: v for k, v in worst.items() if not v < max(GRAD_TOL, k_noise * noise.get(k, 0.0))}
    assert not bad, bad
    assert not hasattr(net, "backward_retries")                  # one pass: nothing was detected-and-repeated


@pytest.mark.parametrize("C,dim_head,window,H,W,modes,n_valid", [
    (64, 16, 6, 12, 18, [1, 0, 1], 3),       # window 6, dim_head 16: 36 tokens per window, 4 heads
    (128, 64, 4, 8, 16, [0, 1, 0], 2),       # dim_head 64 on a tuned window, one padded agent
    (64, 32, 2, 8, 12, [1, 1, 0], 3),        # window 2
])
def test_generic_window_and_dim_head_train(C, dim_head, window, H, W, modes, n_valid):
    """The reference trains every window_size / dim_head its config accepts (hetero_fusion.py:285-327, 82-109).  Shapes outside the
    tuned kernels' window 4 / 8 and dim_head 32 train through the generic exact-f32 attention kernels (csrc/attn.hip k_attention_any
    forward with the row log-sum-exp, csrc/train.hip k_attention_any_bwd) between the same Linears / LayerNorm / warp-adjoint
    kernels as every other shape: forward and every gradient against torch.autograd through the oracle, at the usual bounds."""
    L = 3
    cfg = O.make_config(C, window, L, voxel=0.4, downsample=4, dim_head=dim_head)
    sd = O.random_state_dict(cfg, seed=71)
    scene = O.synthetic_scene(L, C, H, W, modes, n_valid=n_valid, seed=72, tx_step=3.0, ty_step=-2.0, yaw_step=0.25)
    gy = torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(73))
    y_ref, gx_ref, gp_ref = _oracle_grads(cfg, sd, scene, gy)
    net = _net(cfg, sd).eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    assert rel_max_err(y.detach().cpu(), y_ref) < 1e-4
    (y * gy.cuda()).sum().backward()
    _check_grads(net, gp_ref, x.grad, gx_ref)
    # and in training mode (dropout masks regenerated by the backward): a step runs and produces finite gradients
    net.train()
    net.zero_grad()
    y = net(scene[0].cuda().requires_grad_(True), *[t.cuda() for t in scene[1:]])
    y.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


@pytest.mark.parametrize("C,w", [(256, 8), (64, 4)])
def test_recompute_saves_memory_and_changes_nothing(C, w):
    """HmvitFusionTrainDesc::recompute (module.train_recompute = 3): the FFN pre-activations and the queries are not kept; the backward
    recomputes them with the forward's own kernels.  Same output, the same gradients (dropout on: the masks are replayed), fewer saved bytes.
    C = 256: the x16 Linear kernels with their weight images; C = 64: the generic GEMM path."""
    import ctypes
    from hmvit_amd import _lib, train as T
    L, H, W, B = 3, 16, 16, 1
    cfg = O.make_config(C, w, L, voxel=0.4, downsample=4)
    sd = O.random_state_dict(cfg, seed=31)
    scene = [t.cuda() for t in O.synthetic_scene(L, C, H, W, [1, 0, 1], seed=32, B=B, tx_step=3.0, ty_step=-2.0)]
    gy = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(33)).cuda()
    outs, peaks = [], []
    for flag in (False, True):
        net = _net(cfg, sd).train()
        net.train_recompute = 3 if flag else 0
        x = scene[0].clone().requires_grad_(True)
        torch.manual_seed(77)                      # the same dropout stream in both runs
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = net(x, *scene[1:])
        (y * gy).sum().backward()
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
        outs.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    (y0, gx0, gp0), (y1, gx1, gp1) = outs
    # the recomputed rows are the forward's, bit for bit; the gradients agree to the run-to-run level of their atomic sums
    assert torch.equal(y0, y1)
    assert rel_max_err(gx1.cpu(), gx0.cpu()) < 2e-6
    assert gp0.keys() == gp1.keys()
    for k in gp0:
        if float(gp0[k].abs().max()) > 0:
            assert rel_max_err(gp1[k].cpu(), gp0[k].cpu()) < 2e-5, k
    # two (n_slots, P, .) planes less per stage in the saved area (the forward scratch grows by one such plane, transiently)
    mlp, n_stages = cfg["hetero_fusion_block"]["mlp_dim"], 2 * cfg["num_iters"]
    assert peaks[0] - peaks[1] >= (n_stages - 1) * B * L * H * W * (mlp + C) * 4 * 0.9, peaks
