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


def _oracle_grads(cfg, sd, scene, gy, drop_masks=None):
    """Output, d/dx and d/dparam of <oracle(x), gy> by torch.autograd on the CPU."""
    x, pw, mode, rl, mask = scene
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    x = x.clone().requires_grad_(True)
    y = O.hetero_fusion(x, pw, mode, rl, mask, sd, cfg, drop_masks=drop_masks)
    (y * gy).sum().backward()
    return y.detach(), x.grad, {k: v.grad for k, v in sd.items() if v.is_floating_point()}


def _check_grads(net, ref_grads, x_grad, ref_x_grad, used_only=True):
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
    bad = {k: v for k, v in worst.items() if not v < GRAD_TOL}
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
