"""GPU tests of the camera branch's TRAINING path (hm-vit_amd/camera_train.py; VERDICT r2 missing #1): ResNet trunk, cross-view
lift (joint-softmax cross attention with its HIP backward) and up-sampling decoder under ``train()`` against the CPU
restatement of the same modules (oracle/camera_oracle.py, BatchNorm switched to batch statistics) differentiated by
torch.autograd in FLOAT64: outputs, every parameter gradient, the input gradient where there is one, and the running statistics."""
import pytest
import torch

from conftest import rel_max_err
from oracle import camera_oracle as CAM
from oracle import cvt_oracle as CO

pytestmark = pytest.mark.gpu


def _f64(sd):
    return {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in sd.items()}


def _leaf(sd):
    return {k: (v.requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v) for k, v in sd.items()}


def _compare_grads(named_params, ref_sd, prefix_map=lambda k: k, tol=2e-3, report=None):
    gmax = max(float(v.grad.abs().max()) for v in ref_sd.values() if getattr(v, "grad", None) is not None)
    worst = {}
    for name, p in named_params:
        ref = ref_sd[prefix_map(name)]
        if getattr(ref, "grad", None) is None:
            continue
        assert p.grad is not None, name
        scale = max(float(ref.grad.abs().max()), 1e-4 * gmax)
        worst[name] = float((p.grad.cpu().double() - ref.grad).abs().max()) / scale
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"\n{report or 'gradients'}: {len(worst)} parameter tensors, worst", [(k, f"{v:.1e}") for k, v in top])
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, bad
    return worst


def test_cross_attention_backward_kernel():
    """hmvit_cross_attention_train / _backward against torch autograd (float64) of softmax over all cameras' keys."""
    from hmvit_amd.camera_train import CrossAttnFn
    g = torch.Generator().manual_seed(3)
    b, n, Q, K, heads, d = 2, 3, 80, 100, 4, 32          # Q, K not multiples of 64: ragged tiles
    q = torch.randn(b, n, Q, heads * d, generator=g)
    k = torch.randn(b, n, K, heads * d, generator=g)
    v = torch.randn(b, n * K, heads * d, generator=g)
    go = torch.randn(b, Q, heads * d, generator=g)
    qr, kr, vr = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    qh = qr.reshape(b, n, Q, heads, d).permute(0, 3, 1, 2, 4)
    kh = kr.reshape(b, n, K, heads, d).permute(0, 3, 1, 2, 4)
    vh = vr.reshape(b, n * K, heads, d).permute(0, 2, 1, 3)
    dot = d ** -0.5 * torch.einsum("bmnqd,bmnkd->bmnqk", qh, kh).permute(0, 1, 3, 2, 4).reshape(b, heads, Q, n * K)
    ref = torch.einsum("bmqk,bmkd->bmqd", dot.softmax(-1), vh).permute(0, 2, 1, 3).reshape(b, Q, heads * d)
    (ref * go.double()).sum().backward()
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = CrossAttnFn.apply(qc, kc, vc, heads, d)
    (out * go.cuda()).sum().backward()
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 1e-5
    for name, a, r in (("dq", qc, qr), ("dk", kc, kr), ("dv", vc, vr)):
        assert rel_max_err(a.grad.cpu(), r.grad) < 1e-5, name


def test_cvt_camera_encoder_training_matches_float64_autograd():
    """The whole camera branch in training mode (ResNet-18 trunk on 64 x 64 images, two cross-view levels with two Bottlenecks
    each, decoder with two x2 up-samplings): output, parameter gradients and BatchNorm running statistics against the oracle
    restatement with batch-statistics BatchNorm under float64 autograd."""
    from hmvit_amd.camera import CvtCameraEncoder
    ccfg = CAM.make_config(image=64, num_layers=18)
    ccfg["cvm"]["bev_embedding"].update(bev_height=32, bev_width=32)
    csd = CAM.random_state_dict(ccfg, seed=21)
    net = CvtCameraEncoder(ccfg, precision="f32")
    missing, unexpected = net.load_state_dict(csd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing)
    net = net.cuda().train()
    batch = CAM.synthetic_batch(2, ccfg, seed=22)
    ref_sd = _leaf(_f64(csd))
    with CO.batch_statistics():
        ref = CAM.camera_encoder({k: v.double() for k, v in batch.items()}, ref_sd, ccfg)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(23))
    (ref * go.double()).sum().backward()

    out = net({k: v.cuda() for k, v in batch.items()})
    assert out.shape == ref.shape and out.requires_grad
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 1e-4
    (out * go.cuda()).sum().backward()
    # Gradients against float64 autograd.  The branch has 30 ReLUs behind BatchNorms on batch statistics of 8 images; a
    # pre-activation within round-off of zero flips its mask between any two arithmetic orders, and one flip moves the gradients
    # of everything in front of it by up to a per cent (the fp32 run of the oracle itself sits 1e-3 ... 2e-3 from float64 on some
    # trunk tensors, with other flips than the HIP path's; a flip in ResNet layer 3 shows in every tensor of the stem and of
    # layers 1-2, i.e. in almost half of all tensors; one in the decoder shows everywhere).  So the bound is statistical: all
    # gradients taken together (relative L2 over their concatenation) to 2e-2, the median tensor to 2e-2, none beyond 1e-1 - and
    # the single layers are held to tight bounds, where no mask can flip, in test_camera_training_layers_match_float64 below.
    err, num, den = {}, 0.0, 0.0
    nmax = max(float(v.grad.norm()) for v in ref_sd.values() if getattr(v, "grad", None) is not None)
    for name, p in net.named_parameters():
        r = ref_sd[name]
        if getattr(r, "grad", None) is None:
            continue
        assert p.grad is not None, name
        d = (p.grad.cpu().double() - r.grad).norm()
        num, den = num + float(d) ** 2, den + float(r.grad.norm()) ** 2
        # (a convolution bias in front of a batch-statistics BatchNorm has gradient zero in exact arithmetic: absolute floor)
        err[name] = float(d / r.grad.norm().clamp_min(1e-6 * nmax))
    vals = sorted(err.values())
    top = sorted(err.items(), key=lambda kv: -kv[1])[:5]
    total = (num / den) ** 0.5
    print(f"\ncamera branch: {len(err)} parameter gradients vs float64 autograd: all together {total:.1e}; per tensor 40th percentile "
          f"{vals[int(0.4 * len(vals))]:.1e}, median {vals[len(vals) // 2]:.1e}, worst", [(k, f"{v:.1e}") for k, v in top])
    assert len(err) > 150
    # (bounds with room for two or three flips at the very end of the branch, which move EVERY tensor by ~2.4e-3 each - seen on the FAX
    # branch, whose test has the numbers; repeated runs of this one: all together 6e-4 ... 2.7e-3, median 4e-6 ... 3e-4)
    assert total < 2e-2 and vals[len(vals) // 2] < 2e-2 and vals[-1] < 1e-1, (total, top)
    # running statistics moved exactly as nn.BatchNorm2d moves them
    for name, buf in net.named_buffers():
        if "running_" in name:
            assert rel_max_err(buf.cpu(), ref_sd[name]) < 1e-4, name
    # and eval() afterwards serves the folded inference kernels again
    net.eval()
    with torch.no_grad():
        y = net({k: v.cuda() for k, v in batch.items()})
    assert y.shape == ref.shape and not y.requires_grad


def test_camera_training_layers_match_float64():
    """The building blocks of hm-vit_amd/camera_train.py one by one, where no ReLU mask can flip: Linear (ragged K / N), LayerNorm,
    GELU, strided 3 x 3 convolution, strided 1 x 1 convolution and BatchNorm without ReLU - outputs and all gradients against
    torch in float64."""
    import torch.nn.functional as F
    from hmvit_amd import camera_train as CT
    from hmvit_amd import tail_train as TT
    g = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=g)

    def check(name, outs, refs, tol=2e-5):
        for i, (a, r) in enumerate(zip(outs, refs)):
            assert rel_max_err(a.detach().cpu() if a.is_cuda else a, r.detach()) < tol, (name, i)

    # Linear: K = 147 (the unfolded 7 x 7 stem), N = 64, with bias
    x, w, b, gy = rnd(300, 147), rnd(64, 147) * 0.1, rnd(64), rnd(300, 64)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    (F.linear(xr, wr, br) * gy.double()).sum().backward()
    xc, wc, bc = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = CT.LinearFn.apply(xc, wc, bc)
    (y * gy.cuda()).sum().backward()
    check("linear", [y, xc.grad, wc.grad, bc.grad], [F.linear(xr, wr, br), xr.grad, wr.grad, br.grad])
    # LayerNorm(128) and GELU
    x, w, b, gy = rnd(200, 128) * 2 + 0.5, 1 + 0.1 * rnd(128), 0.1 * rnd(128), rnd(200, 128)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    (F.gelu(F.layer_norm(xr, (128,), wr, br, 1e-5)) * gy.double()).sum().backward()
    xc, wc, bc = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = CT.GeluFn.apply(CT.LayerNormFn.apply(xc, wc, bc, 1e-5))
    (y * gy.cuda()).sum().backward()
    check("ln+gelu", [y, xc.grad, wc.grad, bc.grad], [F.gelu(F.layer_norm(xr, (128,), wr, br, 1e-5)), xr.grad, wr.grad, br.grad])
    # 3 x 3 / stride 2 convolution -> BatchNorm (batch statistics, no ReLU) -> strided 1 x 1 convolution, NHWC on the HIP side
    conv = torch.nn.Conv2d(32, 64, 3, 2, 1, bias=False)
    bn = torch.nn.BatchNorm2d(64)
    ds = torch.nn.Conv2d(64, 32, 1, 2, bias=False)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * rnd(64)); bn.bias.copy_(0.1 * rnd(64))
    import copy
    conv_r, bn_r, ds_r = (copy.deepcopy(m).double().train() for m in (conv, bn, ds))
    x, gy = rnd(2, 32, 12, 10), rnd(2, 32, 3, 3)
    xr = x.double().requires_grad_(True)
    ref = ds_r(bn_r(conv_r(xr)))
    (ref * gy.double()).sum().backward()
    conv, bn, ds = conv.cuda().train(), bn.cuda().train(), ds.cuda().train()
    xc = x.cuda().permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    y = CT.conv1x1(TT.bn_relu_module(CT.conv3x3(xc, conv), bn, relu=False), ds, 2)
    (y * gy.cuda().permute(0, 2, 3, 1)).sum().backward()
    check("conv-bn-conv", [y.permute(0, 3, 1, 2), xc.grad.permute(0, 3, 1, 2), conv.weight.grad, bn.weight.grad, bn.bias.grad, ds.weight.grad],
          [ref, xr.grad, conv_r.weight.grad, bn_r.weight.grad, bn_r.bias.grad, ds_r.weight.grad], tol=5e-5)
    assert rel_max_err(bn.running_var.cpu(), bn_r.running_var) < 1e-5 and rel_max_err(bn.running_mean.cpu(), bn_r.running_mean) < 1e-5


def test_fax_camera_encoder_training_matches_float64_autograd():
    """The FAX camera branch in training mode (hm-vit_amd/fax_train.py: ResNet-18 trunk on 64 x 64 images, three
    CrossViewSwapAttention levels - windowed local-to-local and local-to-global cross attention, MLPs -, Bottlenecks, the
    PixelUnshuffle down-sampling blocks, the closing self-attention with its relative-position bias, decoder): output, parameter
    gradients and BatchNorm running statistics against the oracle restatement (oracle/fax_oracle.py) with batch-statistics
    BatchNorm under float64 autograd.  Same statistical bound as the CVT branch (ReLU masks behind tiny-batch BatchNorms)."""
    import hmvit_amd
    from oracle import fax_oracle as FO
    cfg = FO.make_camera_config(image=64)
    cfg["fax"]["self_attn"]["dropout"] = 0.0                 # the one Dropout of the branch: off, so that both sides are deterministic
    torch.manual_seed(9)
    net = hmvit_amd.FaxCameraEncoder(cfg, precision="f32")
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.6, 1.4); m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
    net.set_return_features()
    csd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().train()
    batch = CAM.synthetic_batch(2, CAM.make_config(image=64), seed=10)
    ref_sd = _leaf(_f64(csd))
    with CO.batch_statistics():
        ref = FO.fax_camera_encoder({k: v.double() for k, v in batch.items()}, ref_sd, cfg)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(24))
    (ref * go.double()).sum().backward()

    out = net({k: v.cuda() for k, v in batch.items()})
    assert out.shape == ref.shape and out.requires_grad
    print(f"\nFAX camera branch (training forward): output rel-max {rel_max_err(out.detach().cpu(), ref.detach()):.1e} against float64")
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 1e-4
    (out * go.cuda()).sum().backward()
    err, num, den = {}, 0.0, 0.0
    nmax = max(float(v.grad.norm()) for v in ref_sd.values() if getattr(v, "grad", None) is not None)
    for name, p in net.named_parameters():
        r = ref_sd[name]
        if getattr(r, "grad", None) is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        d = (p.grad.cpu().double() - r.grad).norm()
        num, den = num + float(d) ** 2, den + float(r.grad.norm()) ** 2
        err[name] = float(d / r.grad.norm().clamp_min(1e-6 * nmax))
    vals = sorted(err.values())
    top = sorted(err.items(), key=lambda kv: -kv[1])[:5]
    total = (num / den) ** 0.5
    print(f"\nFAX camera branch: {len(err)} parameter gradients vs float64 autograd: all together {total:.1e}; median "
          f"{vals[len(vals) // 2]:.1e}, worst", [(k, f"{v:.1e}") for k, v in top])
    assert len(err) > 150
    groups = {}
    for k, v in err.items():
        gk = ".".join(k.split(".")[:3]) if k.startswith("fax.") else ".".join(k.split(".")[:2])
        groups.setdefault(gk, []).append(v)
    print("   per group (median):", {k: f"{sorted(v)[len(v) // 2]:.1e}" for k, v in groups.items()})
    # Seen over repeated runs (tools/probe/r03_flaky.sh): every tensor 2.4e-3, 4.2-4.8e-3 or 5.5e-3 from float64 - one, two, ... flipped
    # ReLU masks near the end of the branch (the final map alone holds 11 activations within 1e-5 of zero and disagrees with the
    # float64 forward on 0-1 of its 524 288 masks per run: tools/probe/r03_fax_flips.py); one unit of one channel of the last
    # BatchNorm moves that layer's weight gradient by ~1 / sqrt(2048 x 256) ~ 1.4e-3 and everything in front of it likewise.  A
    # wrong term shows at 0.1-1 on the tensors behind it; the FAX-specific pieces are held to tight bounds one by one in
    # test_fax_training_layers_match_float64, the forward output to 1e-4 above (measured 6.5e-6).
    assert total < 2e-2 and vals[len(vals) // 2] < 2e-2 and vals[-1] < 1e-1, (total, top)
    for name, buf in net.named_buffers():
        if "running_" in name:
            assert rel_max_err(buf.cpu(), ref_sd[name]) < 1e-4, name
    net.eval()
    with torch.no_grad():
        y = net({k: v.cuda() for k, v in batch.items()})
    assert y.shape == ref.shape and not y.requires_grad


def test_fax_training_layers_match_float64():
    """The FAX-specific pieces of hm-vit_amd/fax_train.py one by one against the oracle under float64 autograd: the closing
    self-attention with its relative-position bias (no ReLU: tight), and one CrossViewSwapAttention per kind of level (with and
    without the BEV embedding; the two BatchNorm + ReLU feature projections can flip a mask: 1e-3)."""
    from hmvit_amd.fax import Attention, BEVEmbedding, CrossViewSwapAttention
    from oracle import fax_oracle as FO
    g = torch.Generator().manual_seed(31)
    # self-attention
    att = Attention(128, dim_head=32, dropout=0.0, window_size=8)
    asd = _leaf(_f64({k: v.detach().clone() for k, v in att.state_dict().items()}))
    att = att.cuda().train()
    x = torch.randn(2, 128, 8, 8, generator=g)
    gy = torch.randn(2, 128, 8, 8, generator=g)
    xr = x.double().requires_grad_(True)
    ref = FO.self_attention(xr, asd, 32, 8)
    (ref * gy.double()).sum().backward()
    xc = x.cuda().requires_grad_(True)
    out = att(xc)
    (out * gy.cuda()).sum().backward()
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 2e-5
    assert rel_max_err(xc.grad.cpu(), xr.grad) < 2e-5
    for name, p in att.named_parameters():
        assert rel_max_err(p.grad.cpu(), asd[name].grad) < 2e-5, name
    # the down-sampling block between two levels, against the same torch modules in float64 (train mode)
    import copy
    from hmvit_amd import fax_train as FT
    nn = torch.nn
    seq = nn.Sequential(nn.Conv2d(128, 32, 3, 1, 1, bias=False), nn.PixelUnshuffle(2), nn.Conv2d(128, 128, 3, padding=1, bias=False),
                        nn.BatchNorm2d(128), nn.ReLU(inplace=True), nn.Conv2d(128, 128, 1, padding=0, bias=False), nn.BatchNorm2d(128))
    with torch.no_grad():
        for m in seq.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.normal_(1, 0.1, generator=g); m.bias.normal_(0, 0.1, generator=g)
    ref_seq = copy.deepcopy(seq).double().train()
    seq = seq.cuda().train()
    x = torch.randn(3, 128, 16, 16, generator=g)
    gy = torch.randn(3, 128, 8, 8, generator=g)
    xr = x.double().requires_grad_(True)
    ref = ref_seq(xr)
    (ref * gy.double()).sum().backward()
    xc = x.cuda().requires_grad_(True)
    out = FT.downsample_forward(seq, xc.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    (out * gy.cuda()).sum().backward()
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 2e-5
    assert rel_max_err(xc.grad.cpu(), xr.grad) < 1e-3
    for (name, p), (_, r) in zip(seq.named_parameters(), ref_seq.named_parameters()):
        assert rel_max_err(p.grad.cpu(), r.grad) < 1e-3, name
    for (name, a), (_, r) in zip(seq.named_buffers(), ref_seq.named_buffers()):
        if a.is_floating_point():
            assert rel_max_err(a.cpu(), r) < 1e-5, name
    # CrossViewSwapAttention, level 0 (BEV embedding) and level 1 (plain queries)
    cfg = FO.make_swap_config(64)
    for index in (0, 1):
        fh, H = ((8, 16), (4, 8))[index]
        net = CrossViewSwapAttention(fh, fh, 64, 128, index, **{k: cfg[k] for k in (
            "image_height", "image_width", "no_image_features", "skip", "heads", "dim_head", "qkv_bias", "rel_pos_emb", "q_win_size",
            "feat_win_size", "bev_embedding_flag")})
        sd = FO.swap_state_dict(64, 128, cfg, index, seed=171 + index)
        net.load_state_dict(sd, strict=False)
        ref_sd = _leaf(_f64(sd))
        net = net.cuda().train()
        bev = BEVEmbedding(128, 1.0, 32, 32, 50.0, 50.0, 0.0, [2, 4]).cuda()
        x, feat, I_inv, E_inv = FO.synthetic_inputs(2, 3, 64, fh, fh, 128, H, H, seed=173 + index, image=64)
        gy = torch.randn(2, 128, H, H, generator=g)
        xr, fr = x.double().requires_grad_(True), feat.double().requires_grad_(True)
        with CO.batch_statistics():
            ref = FO.cross_view_swap_attention(xr, getattr(bev, "grid%d" % index).cpu().double(), fr, I_inv.double(), E_inv.double(),
                                               ref_sd, cfg, index)
        (ref * gy.double()).sum().backward()
        xc, fc = x.cuda().requires_grad_(True), feat.cuda().requires_grad_(True)
        out = net(index, xc, bev, fc, I_inv.cuda(), E_inv.cuda())
        (out * gy.cuda()).sum().backward()
        assert rel_max_err(out.detach().cpu(), ref.detach()) < 2e-5, index
        assert rel_max_err(xc.grad.cpu(), xr.grad) < 1e-3 and rel_max_err(fc.grad.cpu(), fr.grad) < 1e-3, index
        n = 0
        # (the LayerNorm bias / Linear bias in front of the keys has a gradient that is zero in exact arithmetic - a constant added
        # to every key shifts every logit of a row alike -: such tensors are held to 1e-4 of the module's largest gradient)
        gmax = max(float(v.grad.abs().max()) for v in ref_sd.values() if getattr(v, "grad", None) is not None)
        for name, p in net.named_parameters():
            r = ref_sd[name]
            if getattr(r, "grad", None) is None:
                continue
            n += 1
            e = float((p.grad.cpu().double() - r.grad).abs().max() / max(float(r.grad.abs().max()), 1e-4 * gmax))
            assert e < 1e-3, (index, name, e)
        assert n > 30


def test_attention_with_bias_operator_matches_float64_autograd():
    """hmvit_attention_bias_train / _backward (the FAX self-attention's products) against torch float64: output,
    dq / dk / dv and the (heads, Q, K) bias gradient summed over the batch; Q = K = 100 (not a multiple of the 64-row tiles)."""
    from hmvit_amd import camera_train as CT
    g = torch.Generator().manual_seed(3)
    b, N, m, dh = 3, 100, 4, 32
    q, k, v = (torch.randn(b, N, m * dh, generator=g) for _ in range(3))
    bias = torch.randn(m, N, N, generator=g)
    da = torch.randn(b, N, m * dh, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (q, k, v, bias)]
    qh, kh, vh = (t.reshape(b, N, m, dh).permute(0, 2, 1, 3) for t in ref_in[:3])
    sim = qh @ kh.transpose(-1, -2) * dh ** -0.5 + ref_in[3][None]
    ref = (sim.softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(b, N, m * dh)
    (ref * da.double()).sum().backward()
    ins = [t.cuda().requires_grad_(True) for t in (q, k, v, bias)]
    out = CT.AttnBiasFn.apply(*ins, m, dh)
    (out * da.cuda()).sum().backward()
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 2e-6
    for name, a, r in zip("q k v bias".split(), ins, ref_in):
        assert rel_max_err(a.grad.cpu(), r.grad) < 5e-6, name


def test_maxpool_operator_matches_torch_including_ties():
    """hmvit_maxpool2d + hmvit_maxpool2d_backward (3 x 3 / stride 2 / pad 1, NHWC f32) against F.max_pool2d in float64 - on a map with
    many exact ties (post-ReLU zeros), where the gradient must go to the first maximum of the window in row-major order."""
    import torch.nn.functional as F
    from hmvit_amd import camera_train as CT
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(2, 17, 22, 16, generator=g))            # NHWC, odd sizes, ~half zeros
    x[0, 4:9, 3:8] = 1.25                                              # a plateau: every window inside it is all ties
    dy = torch.randn(2, 9, 11, 16, generator=g)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    (yr * dy.double().permute(0, 3, 1, 2)).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = CT.MaxPoolFn.apply(xg, 3, 2, 1)
    (y * dy.cuda()).sum().backward()
    assert torch.equal(y.detach().cpu().double(), yr.detach().permute(0, 2, 3, 1))
    assert float((xg.grad.cpu().double() - xr.grad.permute(0, 2, 3, 1)).abs().max()) < 1e-6


def _replayed_mask_check(monkeypatch, net, batch, oracle_forward, csd, tag, bound=1e-4):
    """HIP training forward + backward with every ReLU mask recorded, the float64 restatement with those masks replayed, every
    parameter gradient compared (relative L2 per tensor)."""
    import torch.nn.functional as F
    from hmvit_amd import tail_train as TT
    tape = []
    real_bn_relu, real_relu, real_frelu = TT.bn_relu_module, torch.relu, F.relu

    def rec_bn_relu(x, bn, relu=True):
        y = real_bn_relu(x, bn, relu)
        if relu:
            tape.append((y.detach() > 0).permute(0, 3, 1, 2).cpu())      # NHWC -> the oracle's NCHW
        return y

    def rec_relu(x):
        y = real_relu(x)
        tape.append((y.detach() > 0).permute(0, 3, 1, 2).cpu())
        return y

    monkeypatch.setattr(TT, "bn_relu_module", rec_bn_relu)
    monkeypatch.setattr(torch, "relu", rec_relu)
    out = net({k: v.cuda() for k, v in batch.items()})
    monkeypatch.setattr(TT, "bn_relu_module", real_bn_relu)
    monkeypatch.setattr(torch, "relu", real_relu)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(43))
    (out * go.cuda()).sum().backward()
    assert len(tape) > 30
    cursor = [0]

    def replay(x, inplace=False):
        m = tape[cursor[0]]
        cursor[0] += 1
        assert tuple(m.shape) == tuple(x.shape), (cursor[0], m.shape, x.shape)
        return x * m.to(x.dtype)

    ref_sd = _leaf(_f64(csd))
    monkeypatch.setattr(F, "relu", replay)
    with CO.batch_statistics():
        ref = oracle_forward({k: v.double() for k, v in batch.items()}, ref_sd)
    monkeypatch.setattr(F, "relu", real_frelu)
    assert cursor[0] == len(tape), (cursor[0], len(tape))
    assert rel_max_err(out.detach().cpu(), ref.detach()) < 1e-4
    (ref * go.double()).sum().backward()
    nmax = max(float(v.grad.norm()) for v in ref_sd.values() if getattr(v, "grad", None) is not None)
    err = {}
    for name, p in net.named_parameters():
        r = ref_sd[name]
        if getattr(r, "grad", None) is None:
            continue
        assert p.grad is not None, name
        err[name] = float((p.grad.cpu().double() - r.grad).norm() / r.grad.norm().clamp_min(1e-2 * nmax))
    top = sorted(err.items(), key=lambda kv: -kv[1])[:5]
    print(f"\n{tag} with replayed ReLU masks: {len(err)} parameter gradients vs float64 autograd, worst", [(k, f"{v:.1e}") for k, v in top])
    assert len(err) > 150
    assert top[0][1] < bound, top


def test_cvt_branch_every_parameter_gradient_with_the_relu_masks_replayed(monkeypatch):
    """the float64 comparison of the whole branch above can only be statistical, because ReLU masks behind
    8-image BatchNorms flip at round-off and one flip moves half of all tensors by a per cent - so a wiring error between two layers
    that stays under 10 % on one tensor could hide in its bounds.  Here the flips are taken out of the comparison: the HIP run records
    the mask of every ReLU it applies (the fused BatchNorm + ReLU outputs and the residual ReLUs, in call order), and the float64
    restatement replays them (`x * mask` instead of `relu(x)`), so both sides differentiate the SAME piecewise-linear function.  Every
    parameter gradient is then held to 1e-4 (relative L2 per tensor, measured 9e-6; tensors whose gradient is zero in exact arithmetic
    - a convolution bias in front of a batch-statistics BatchNorm - against 1e-2 of the largest norm)."""
    from hmvit_amd.camera import CvtCameraEncoder
    ccfg = CAM.make_config(image=64, num_layers=18)
    ccfg["cvm"]["bev_embedding"].update(bev_height=32, bev_width=32)
    csd = CAM.random_state_dict(ccfg, seed=41)
    net = CvtCameraEncoder(ccfg, precision="f32")
    net.load_state_dict(csd, strict=False)
    net = net.cuda().train()
    batch = CAM.synthetic_batch(2, ccfg, seed=42)
    _replayed_mask_check(monkeypatch, net, batch, lambda b, sd: CAM.camera_encoder(b, sd, ccfg), csd, "CVT camera branch")


def test_fax_branch_every_parameter_gradient_with_the_relu_masks_replayed(monkeypatch):
    """The same for the FAX camera branch (hm-vit_amd/fax_train.py against oracle/fax_oracle.py), self-attention dropout off."""
    import hmvit_amd
    from oracle import fax_oracle as FO
    cfg = FO.make_camera_config(image=64)
    cfg["fax"]["self_attn"]["dropout"] = 0.0
    torch.manual_seed(19)
    net = hmvit_amd.FaxCameraEncoder(cfg, precision="f32")
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.6, 1.4); m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
    net.set_return_features()
    csd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda().train()
    batch = CAM.synthetic_batch(2, CAM.make_config(image=64), seed=20)
    _replayed_mask_check(monkeypatch, net, batch, lambda b, sd: FO.fax_camera_encoder(b, sd, cfg), csd, "FAX camera branch")
