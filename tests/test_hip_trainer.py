"""GPU test of the train harness (hm-vit_amd/trainer.py): the reference's per-batch loop on synthetic replay scenes
with the fusion's HIP forward + backward, a frozen HIP LiDAR encoder, checkpoints and resume."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Args:
    grid, agents, small, precision, seed, frames, train_lidar_backbone = [128, 96], 3, True, "f32", 0, 4, False
    val_frames, camera_ratio, ego_mode, camera_image, train_camera_backbone = 0, 0.0, "mixed", 64, False


def test_train_loop_learns_saves_and_resumes(tmp_path):
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T
    hypes = T.default_hypes(epoches=3)
    cfg, model, pre, post, ds, _ = T.build(_Args)
    model = model.cuda()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = T.train(model, ds, pre, hypes, saved_path=str(tmp_path))
    assert len(res["epoch_loss"]) == 3 and all(l == l for l in res["epoch_loss"])          # finite
    assert res["epoch_loss"][-1] < res["epoch_loss"][0], res                                  # the loop optimises
    after = model.state_dict()
    moved = lambda k: float((after[k].float() - before[k].float()).abs().max())
    # the fusion (HIP backward) and the tail (torch modules) moved; the frozen encoder and the unused heads did not
    assert moved("fusion_net.hetero_fusion_block.window_attention.q_linears.1.weight") > 0
    assert moved("fusion_net.hetero_fusion_block.window_attention.relation_att") > 0
    assert moved("fusion_net.mlp_head.net.1.0.weight") > 0
    assert moved("decoder.lidar_cls_head.weight") > 0
    assert all(moved(k) == 0 for k in before if k.startswith("lidar_encoder.") and "num_batches" not in k)
    assert moved("fusion_net.hetero_fusion_block.window_attention.q_linears.0.weight") == 0     # camera-type weights: no agent
    assert moved("cls_head.weight") == 0
    assert sorted(os.listdir(tmp_path)) == ["net_epoch1.pth", "net_epoch2.pth", "net_epoch3.pth"]
    # resume: a fresh model picks up epoch 3's weights (train_utils.py:40-75)
    _, fresh, _, _, _, _ = T.build(_Args)
    epoch, fresh = T.load_saved_model(str(tmp_path), fresh)
    assert epoch == 3
    for k, v in fresh.state_dict().items():
        assert torch.equal(v.cpu(), after[k].cpu()), k
    # evaluation after training runs on the HIP inference kernels (BatchNorm folded) and agrees with the torch modules in eval mode
    model.eval()
    batch = T.to_batch(ds[0], pre, "cuda")
    with torch.no_grad():
        out = model(batch)
        x, mask = hmvit_amd.model.regroup(model.lidar_encoder(model._lidar_batch(batch, torch.ones(3, dtype=torch.int))), [3], 3)
        fused = model.fusion_net(x, batch["pairwise_t_matrix"], batch["mode"].int(), batch["record_len"], mask)
        ref_psm, ref_rm = model.decoder._forward_training(fused.unsqueeze(1), torch.ones(1, 3, dtype=torch.int), torch_modules=True)
    assert float((out["psm"] - ref_psm).abs().max() / ref_psm.abs().max()) < 1e-4
    assert float((out["rm"] - ref_rm).abs().max() / ref_rm.abs().max()) < 1e-4


def _tail_reference(dec, x, modes):
    """hetero_decoder.py:55-89 with use_upsample=False on the module's own layers (any device / dtype)."""
    ego = [m[0] for m in modes]
    psm, rm = [None] * len(ego), [None] * len(ego)
    for v, name in ((0, "camera"), (1, "lidar")):
        idx = [b for b, e in enumerate(ego) if e == v]
        if not idx:
            continue
        t = x[idx, 0]
        for layer in getattr(dec, f"{name}_decoder").decoder:
            t = layer(t)
        p, r = getattr(dec, f"{name}_cls_head")(t), getattr(dec, f"{name}_reg_head")(t)
        for k, b in enumerate(idx):
            psm[b], rm[b] = p[k], r[k]
    return torch.stack(psm), torch.stack(rm)


@pytest.mark.parametrize("gscale", [1.0, 1e-6])
@pytest.mark.parametrize("modes", [[[1, 1], [1, 0]], [[0, 1], [1, 1], [0, 0]]])
def test_tail_training_kernels_match_torch_modules(modes, gscale):
    """The detection tail in training mode on libhmvit (conv3x3 / BatchNorm on batch statistics + ReLU / 1x1 heads, forward and
    backward, hm-vit_amd/tail_train.py) against the same torch modules under torch autograd: outputs, input gradient, every
    parameter gradient and the running statistics.  gscale 1e-6: the upstream gradient at the magnitude a normalised focal loss
    hands back - the split-f16 products of the backward must not lose their low halves there (_lib.grad_pow2)."""
    import copy
    import hmvit_amd
    from oracle import decoder_oracle as DO
    torch.manual_seed(5)
    params = DO.make_params()
    net = hmvit_amd.HeteroDecoder(params, precision="split")
    net.load_state_dict(DO.random_state_dict(params, 31), strict=True)
    # reference: the same torch modules in float64 on the CPU (the GPU library convolutions behind torch are free to pick
    # reduced-precision algorithms, which made a float32 GPU reference flaky at the 1e-3 level)
    ref = copy.deepcopy(net).double().train()
    net = net.cuda().train()
    B = len(modes)
    mode = torch.tensor(modes)
    x = torch.randn(B, 1, 256, 12, 10)
    gp, gr = torch.randn(B, 2, 12, 10) * gscale, torch.randn(B, 14, 12, 10) * gscale
    outs = []
    for m, torch_modules, conv in ((net, False, lambda t: t.cuda()), (ref, True, lambda t: t.double())):
        xi = conv(x).clone().requires_grad_(True)
        if torch_modules:
            psm, rm = _tail_reference(m, xi, modes)
        else:
            psm, rm = m._forward_training(xi, mode)
        ((psm * conv(gp)).sum() + (rm * conv(gr)).sum()).backward()
        outs.append((psm.detach().cpu().double(), rm.detach().cpu().double(), xi.grad.cpu().double()))
    err = lambda a, b: float((a.cpu().double() - b.cpu().double()).abs().max() / b.abs().max().clamp_min(1e-12))
    assert err(outs[0][0], outs[1][0]) < 1e-4 and err(outs[0][1], outs[1][1]) < 1e-4

    def close(a, b, tol, floor=0.0):
        """Gradients pass through ReLU masks: an activation within round-off of zero (about one in a million, and the batch
        statistics are summed with atomics, so which one varies from run to run) flips its mask on one side, and the flip of one
        unit of a late layer reaches, through the 3x3 convolutions before it, thousands of entries of an early gradient at the
        1e-3 level (seen: 8.5e-4 of the largest entry).  A wrong tap or a missing term shows at the 0.1-1 level, so the bound
        is: rms error < `tol`, no entry further than 20 `tol`, relative to the largest entry (or to `floor`, for tensors that
        are zero in exact arithmetic)."""
        a, b = a.cpu().double(), b.cpu().double()
        scale = max(float(b.abs().max()), floor, 1e-12)
        d = (a - b).abs() / scale
        return float(d.pow(2).mean().sqrt()) < tol and float(d.max()) < 20 * tol
    assert close(outs[0][2], outs[1][2], 3e-4)
    used = 0
    # a convolution bias in front of a BatchNorm has a gradient that is zero in exact arithmetic (the batch mean absorbs it):
    # round-off noise on both sides, held to 1e-3 of the model's largest gradient instead of to its own magnitude
    gmax = max(float(q.grad.abs().max()) for q in ref.parameters() if q.grad is not None)
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, k
            continue
        used += 1
        assert close(p.grad, q.grad, 5e-4, floor=1e-3 * gmax), k
    assert used >= 12
    for (k, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        assert err(a, b) < 1e-5 if a.is_floating_point() else bool((a.cpu() == b).all()), k


def _pointpillar_reference(net, vf, vc, vn, n_agents):
    """PointPillar.forward (features) from the module's own torch layers, any device / dtype: pillar_vfe.py:105-146,31-53,
    point_pillar_scatter.py:14-47, base_bev_backbone.py:89-122, downsample_conv.py:32-51."""
    from hmvit_amd.encoder_train import pfn_features
    nx, ny, _ = [int(v) for v in net.scatter_cfg["grid_size"]]
    pfn = net.pillar_vfe.pfn_layers[0]
    feats = pfn_features(vf, vc.long(), vn, net.args["voxel_size"], net.args["lidar_range"])
    h = pfn.linear(feats)
    h = torch.relu(pfn.norm(h.permute(0, 2, 1)).permute(0, 2, 1)).max(dim=1)[0]
    canvas = torch.zeros(n_agents, 64, ny * nx, dtype=h.dtype)
    for b in range(n_agents):
        m = vc[:, 0] == b
        canvas[b][:, (vc[m, 2] * nx + vc[m, 3]).long()] = h[m].t()
    x = canvas.view(n_agents, 64, ny, nx)
    ups = []
    for blk, de in zip(net.backbone.blocks, net.backbone.deblocks):
        x = blk(x)
        ups.append(de(x))
    x = torch.cat(ups, dim=1)
    for dc in net.shrink_conv.layers:
        x = dc.double_conv(x)
    return x


def test_pointpillar_training_matches_torch_modules():
    """The LiDAR encoder in training mode on libhmvit (hm-vit_amd/encoder_train.py: PFN Linear + BatchNorm1d, strided 3x3
    convolutions, transposed convolutions, BatchNorm2d on batch statistics, shrink header - forward and backward) against its own
    torch layers in float64 on the CPU: features, every parameter gradient, running statistics."""
    import copy
    import hmvit_amd
    from oracle import pointpillar_oracle as PO
    args = PO.make_args(64, 48, small=True)
    sd = PO.random_state_dict(args, seed=3)
    vf, vc, vn = PO.synthetic_pillars(2, 400, 64, 48, args, seed=4)
    net = hmvit_amd.PointPillar(args, precision="split")
    net.load_state_dict(sd, strict=True)
    net.set_return_features()
    ref = copy.deepcopy(net).double().train()
    ref32 = copy.deepcopy(net).float().train()          # torch's own float32 run: the yardstick for what float32 can hold here
    net = net.cuda().train()
    y = net({"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}, "n_agents": 2})
    y_ref = _pointpillar_reference(ref, vf.double(), vc, vn, 2)
    y_32 = _pointpillar_reference(ref32, vf.float(), vc, vn, 2)
    assert y.shape == y_ref.shape
    g = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(1))
    (y * g.cuda()).sum().backward()
    (y_ref * g.double()).sum().backward()
    (y_32 * g).sum().backward()
    err = lambda a, b: float((a.cpu().double() - b.cpu().double()).abs().max() / b.abs().max().clamp_min(1e-12))
    assert err(y.detach(), y_ref.detach()) < 1e-4

    gmax = max(float(q.grad.abs().max()) for q in ref.parameters() if q.grad is not None)
    used, worst = 0, {}

    def rel(a, b, floor):
        """(rms error, entries off by more than 4e-3 as a multiple of the allowance max(2, numel / 1000)), both relative to the tensor's
        largest entry (or `floor`)"""
        a, b = a.cpu().double(), b.cpu().double()
        d = (a - b).abs() / max(float(b.abs().max()), floor, 1e-12)
        return float(d.pow(2).mean().sqrt()), float((d > 4e-3).double().sum()) / max(2.0, 1e-3 * d.numel())   # > 1: too many
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, k
            continue
        used += 1
        worst[k] = rel(p.grad, q.grad, 1e-3 * gmax)
    # Ten layers of tiny-batch BatchNorm backward, ReLU and max masks: torch's own float32 run is 1-1.5e-3 (rms, relative to the
    # tensor's largest entry) away from float64 on the deepest parameters.  The HIP path is held to that yardstick: no tensor
    # more than 3x further from float64 than torch-float32 is (floor 2e-4), at most max(2, numel / 1000) entries off by 4e-3.
    yard = {k: rel(p.grad, q.grad, 1e-3 * gmax)[0] for (k, p), (_, q) in zip(ref32.named_parameters(), ref.named_parameters())
            if q.grad is not None}
    bad = {k: (v, yard[k]) for k, v in worst.items() if not (v[0] < max(3 * yard[k], 2e-4) and v[1] <= 1.0)}
    assert not bad, bad
    assert used >= 30
    for (k, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        if a.is_floating_point():
            assert err(a, b) < 1e-4, k
    # eval() goes back to the folded inference kernels and sees the updated running statistics
    net.eval(); ref.eval()
    with torch.no_grad():
        ye = net({"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}, "n_agents": 2})
        ye_ref = _pointpillar_reference(ref, vf.double(), vc, vn, 2)
    assert err(ye, ye_ref) < 2e-4


def test_whole_model_trains_with_unfrozen_encoder():
    """train_camera.py without --fix_lidar_backbone: encoder, fusion and detection tail all on the tape; one AdamW step moves
    parameters of all three."""
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T

    class A(_Args):
        train_lidar_backbone = True
    hypes = T.default_hypes(epoches=1)
    cfg, model, pre, post, ds, _ = T.build(A)
    model = model.cuda()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = T.train(model, ds, pre, hypes)
    assert res["epoch_loss"][0] == res["epoch_loss"][0]
    after = model.state_dict()
    moved = lambda k: float((after[k].float() - before[k].float()).abs().max())
    assert moved("lidar_encoder.pillar_vfe.pfn_layers.0.linear.weight") > 0
    assert moved("lidar_encoder.backbone.blocks.0.1.weight") > 0 and moved("lidar_encoder.backbone.deblocks.2.0.weight") > 0
    assert moved("lidar_encoder.shrink_conv.layers.0.double_conv.2.bias") > 0
    assert moved("fusion_net.hetero_fusion_block.grid_attention.relation_msg") > 0 and moved("decoder.lidar_reg_head.weight") > 0


def test_two_rank_ddp_training_on_one_gpu():
    """The train loop under torch.distributed.run with two ranks (sharded frames, DistributedDataParallel buckets over the custom
    autograd Functions, unused parameters) - both ranks share this box's GPU, so the process group is gloo; on a multi-GPU node the
    same command with --backend nccl is the RCCL run."""
    import json
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "hmvit_amd.trainer", "--epochs", "2", "--frames", "4", "--agents", "3", "--grid", "128", "96",
           "--small", "--backend", "gloo"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["world_size"] == 2 and res["steps"] >= 3
    assert all(l == l for l in res["epoch_loss"]) and res["epoch_loss"][-1] < res["epoch_loss"][0]
    assert res["rank_param_spread"] <= 1e-6, res["rank_param_spread"]     # DDP left both ranks with the same parameters


def test_mixed_camera_lidar_five_agent_train_loop_with_validation():
    """BASELINE configs[4] in one process: 5 agents per scene, modality rolled per agent (camera_to_lidar_ratio 0.5, ego_mode
    mixed, basedataset.py:193-200), the batch of mixed/intermediate_fusion_dataset.py:398-415 (camera / intrinsic / extrinsic next
    to processed_lidar), CVT lift in the camera slot (frozen, --fix_camera_backbone), frozen PointPillar, the fusion and the tail
    training on the HIP backward kernels, a validation pass every epoch (train_camera.py:201-220) that flips eval() / train()."""
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T

    class A(_Args):
        agents, frames, val_frames, camera_ratio, seed = 5, 4, 2, 0.5, 1      # seed 1: the four frames hold camera AND LiDAR egos
    hypes = T.default_hypes(epoches=3)
    cfg, model, pre, post, ds, val = T.build(A)
    rolled = [ds.roll_modes(i) for i in range(len(ds))]
    assert any(0 in m for m in rolled) and any(1 in m for m in rolled) and any(m[0] == 0 for m in rolled) and any(m[0] == 1 for m in rolled), rolled
    frame = ds[0]
    assert frame["camera"].shape == (5, 4, 64, 64, 3) and frame["intrinsic"].shape == (5, 4, 3, 3) and frame["extrinsic"].shape == (5, 4, 4, 4)
    model = model.cuda()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = T.train(model, ds, pre, hypes, val_dataset=val)
    assert len(res["epoch_loss"]) == 3 and len(res["val_loss"]) == 3
    assert all(l == l for l in res["epoch_loss"] + res["val_loss"])                          # finite
    assert res["epoch_loss"][-1] < res["epoch_loss"][0], res
    after = model.state_dict()
    moved = lambda k: float((after[k].float() - before[k].float()).abs().max())
    # both agent types were in the scenes: the typed parameters of BOTH types moved (fusion on the HIP backward)
    for t in (0, 1):
        assert moved(f"fusion_net.hetero_fusion_block.window_attention.q_linears.{t}.weight") > 0
        assert moved(f"fusion_net.hetero_fusion_block.grid_ffd.fn.net.{t}.0.weight") > 0
    assert moved("fusion_net.hetero_fusion_block.grid_attention.relation_att") > 0
    # camera and LiDAR egos both occurred: both typed decoders and both mlp_head branches trained
    assert moved("decoder.lidar_cls_head.weight") > 0 and moved("decoder.camera_cls_head.weight") > 0
    assert moved("fusion_net.mlp_head.net.0.0.weight") > 0 and moved("fusion_net.mlp_head.net.1.0.weight") > 0
    # frozen encoders did not move
    assert all(moved(k) == 0 for k in before if k.startswith(("lidar_encoder.", "camera_encoder.")) and "num_batches" not in k)
    assert not model.training                  # validate() ran last: the model is left in eval mode, as in the reference loop


def test_mixed_batch_trains_the_camera_encoder_too():
    """train_camera.py without --fix_camera_backbone on a mixed batch: the CVT camera encoder (ResNet trunk, cross-view lift,
    decoder) is on the tape next to the fusion and the tail; one epoch moves parameters of every part of it and its BatchNorm
    running statistics, and the loss stays finite."""
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T

    class A(_Args):
        agents, frames, val_frames, camera_ratio, seed, train_camera_backbone = 3, 3, 1, 0.6, 1, True
    hypes = T.default_hypes(epoches=1)
    cfg, model, pre, post, ds, val = T.build(A)
    assert any(0 in ds.roll_modes(i) for i in range(len(ds)))
    model = model.cuda()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = T.train(model, ds, pre, hypes, val_dataset=val)
    assert all(l == l for l in res["epoch_loss"] + res["val_loss"])
    after = model.state_dict()
    moved = lambda k: float((after[k].float() - before[k].float()).abs().max())
    for k in ("camera_encoder.encoder.encoder.conv1.weight", "camera_encoder.encoder.encoder.layer2.0.conv1.weight",
              "camera_encoder.encoder.encoder.layer4.1.bn2.weight", "camera_encoder.cvm.cross_views.0.cross_attend.to_q.1.weight",
              "camera_encoder.cvm.cross_views.1.img_embed.weight", "camera_encoder.cvm.bev_embedding.learned_features",
              "camera_encoder.cvm.layers.0.0.conv2.weight", "camera_encoder.decoder.decoder.0.weight",
              "camera_encoder.encoder.encoder.bn1.running_mean", "fusion_net.hetero_fusion_block.window_attention.q_linears.0.weight"):
        assert moved(k) > 0, k
