"""HIP camera -> BEV lift (hm-vit_amd/cvt.py, csrc/cvt.hip) against the golden run of the reference's CrossViewAttention
(tests/golden/g11_cross_view.npz) and the oracle at the shipped level-2 size."""
import pytest
import torch

from conftest import load_golden, rel_max_err
from oracle import cvt_oracle as CO

pytestmark = pytest.mark.gpu
TOL = 1e-4   # f32 kernels


def _net(feat_hw, feat_dim, dim, cfg, sd):
    from hmvit_amd.cvt import CrossViewAttention
    net = CrossViewAttention(feat_hw, feat_hw, feat_dim, dim, cfg)
    if cfg["no_image_features"]:
        sd = {k: v for k, v in sd.items() if not k.startswith("feature_proj.")}     # the module has no such branch then
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    return net.cuda().eval()


@pytest.mark.parametrize("tag,no_feat,skip", [("a", False, True), ("b", True, False)])
def test_cross_view_attention_golden(tag, no_feat, skip):
    from hmvit_amd.cvt import BEVEmbedding
    g = load_golden("g11_cross_view.npz")
    cfg = CO.make_config()
    cfg["no_image_features"], cfg["skip"] = no_feat, skip
    sd = CO.random_state_dict(64, 128, cfg, seed=int(g["seed_weights"]))
    net = _net(12, 64, 128, cfg, sd)
    bev = BEVEmbedding(128, 1.0, 64, 64, 100.0, 100.0, 0.0, [128, 128, 64]).cuda()
    assert torch.allclose(bev.grid.cpu(), CO.bev_grid(64, 64, 100.0, 100.0, 0.0, 3))
    x, feat, I_inv, E_inv = CO.synthetic_inputs(2, 4, 64, 12, 12, 128, 8, 8, seed=int(g["seed_inputs"]))
    y = net(x.cuda(), bev, feat.cuda(), I_inv.cuda(), E_inv.cuda()).cpu()
    assert y.shape == g["y_" + tag].shape
    assert rel_max_err(y, g["y_" + tag]) < TOL


@pytest.mark.parametrize("precision,tol", [("f32", TOL), ("f16", 1e-2)])
def test_cross_view_attention_level2_size_vs_oracle(precision, tol):
    """Second pyramid level of the shipped CVT config: (4, 512, 16, 16) features, 32 x 32 BEV queries, dim 128, 4 heads."""
    from hmvit_amd.cvt import BEVEmbedding
    cfg = CO.make_config()
    sd = CO.random_state_dict(512, 128, cfg, seed=5)
    net = _net(16, 512, 128, cfg, sd)
    net.cross_attend.precision = precision
    bev = BEVEmbedding(128, 1.0, 256, 256, 100.0, 100.0, 0.0, [128, 128, 64]).cuda()
    x, feat, I_inv, E_inv = CO.synthetic_inputs(1, 4, 512, 16, 16, 128, 32, 32, seed=6)
    ref = CO.cross_view_attention(x, bev.grid.cpu(), feat, I_inv, E_inv, sd, cfg)
    y = net(x.cuda(), bev, feat.cuda(), I_inv.cuda(), E_inv.cuda()).cpu()
    assert rel_max_err(y, ref) < tol
    with pytest.raises(RuntimeError):
        net(x, bev, feat, I_inv, E_inv)          # CPU tensors: no fallback


@pytest.mark.parametrize("b,n,Q,K,heads", [(1, 1, 64, 64, 1), (2, 3, 128, 192, 4), (1, 4, 1024, 1024, 4)])
def test_cross_attention_core_f16_vs_torch(b, n, Q, K, heads):
    """Matrix-core attention core against a torch f32 evaluation of the same f16 operands (joint softmax over the cameras'
    keys, cvt_modules.py:148-158); logits spread over several units so the running-max rescale is exercised."""
    import ctypes
    from hmvit_amd import _lib
    gen = torch.Generator().manual_seed(b * 1000 + Q + K)
    hd = heads * 32
    q = (torch.randn(b, n, Q, hd, generator=gen) * 1.5).half()
    k = (torch.randn(b, n, K, hd, generator=gen) * 1.5).half()
    v = torch.randn(b, n * K, hd, generator=gen).half()
    qf = q.float().reshape(b, n, Q, heads, 32).permute(0, 3, 1, 2, 4)
    kf = k.float().reshape(b, n, K, heads, 32).permute(0, 3, 1, 2, 4)
    dot = (32 ** -0.5) * torch.einsum("bmnqd,bmnkd->bmnqk", qf, kf)
    att = dot.permute(0, 1, 3, 2, 4).reshape(b, heads, Q, n * K).softmax(-1)
    ref = torch.einsum("bmqk,bkmd->bqmd", att, v.float().reshape(b, n * K, heads, 32)).reshape(b, Q, hd)
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    out = torch.empty(b, Q, hd, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib.hmvit_cross_attention(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), b, n, Q, K, heads, 32,
                                              _lib.PREC_F16, st), "cross_attention")
    assert rel_max_err(out.cpu(), ref) < 3e-3       # P and the scaled q operand are rounded to f16
    assert _lib.lib.hmvit_cross_attention(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), b, n, Q - 1, K, heads, 32,
                                          _lib.PREC_F16, st) != 0     # Q not a multiple of 64: refused, no silent fallback


@pytest.mark.parametrize("b,n,Q,K,heads,spread", [(1, 1, 64, 64, 1, 1.5), (2, 3, 128, 192, 4, 1.5), (1, 4, 1024, 1024, 4, 3.0),
                                                 (2, 2, 64, 200, 2, 1.5)])
def test_cross_attention_core_split_vs_float64(b, n, Q, K, heads, spread):
    """HMVIT_PREC_SPLIT: the split-f16 matrix-core kernel (f32 operands as hi + lo halves, three products each) against a float64
    evaluation of the joint softmax over the cameras' keys (cvt_modules.py:148-158), at fp32 round-off class and no worse than the
    exact-f32 kernel it replaces; logits spread over many units (peaked rows, running-max rescales).  K = 200 is not a multiple
    of 64: that call keeps the exact-f32 kernel."""
    import ctypes
    from hmvit_amd import _lib
    gen = torch.Generator().manual_seed(b * 1000 + Q + K + 1)
    hd = heads * 32
    q = torch.randn(b, n, Q, hd, generator=gen) * spread
    k = torch.randn(b, n, K, hd, generator=gen) * spread
    v = torch.randn(b, n * K, hd, generator=gen) * 2.0
    qd = q.double().reshape(b, n, Q, heads, 32).permute(0, 3, 1, 2, 4)
    kd = k.double().reshape(b, n, K, heads, 32).permute(0, 3, 1, 2, 4)
    dot = (32 ** -0.5) * torch.einsum("bmnqd,bmnkd->bmnqk", qd, kd)
    att = dot.permute(0, 1, 3, 2, 4).reshape(b, heads, Q, n * K).softmax(-1)
    ref = torch.einsum("bmqk,bkmd->bqmd", att, v.double().reshape(b, n * K, heads, 32)).reshape(b, Q, hd)
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    for name, prec in (("split", _lib.PREC_SPLIT), ("f32", _lib.PREC_F32)):
        out = torch.empty(b, Q, hd, device="cuda")
        _lib.check(_lib.lib.hmvit_cross_attention(qc.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), b, n, Q, K, heads, 32,
                                                  prec, st), "cross_attention")
        outs[name] = rel_max_err(out.cpu().double(), ref)
    print(f"cross attention b={b} n={n} Q={Q} K={K}: split {outs['split']:.2e}, exact-f32 kernel {outs['f32']:.2e}")
    assert outs["split"] < 5e-6 and outs["split"] < 4 * outs["f32"] + 1e-6


def test_split_attention_core_is_declined_when_the_projections_could_leave_f16_range():
    """The split-product core has no range normalisation: with key weights 1000 x their usual size the module must fall back to
    the exact-f32 kernel (and still match float64), as the f32 kernel did for such weights before."""
    from hmvit_amd.cvt import CrossAttention, split_linears
    torch.manual_seed(0)
    m = CrossAttention(128, 4, 32, True).cuda().eval()
    assert m._split_core_in_range()
    with torch.no_grad():
        m.to_k[1].weight.mul_(1000.0)
        m.to_q[1].weight.mul_(1e-3)
    assert not m._split_core_in_range()
    q, k, v = torch.randn(1, 2, 64, 128).cuda(), torch.randn(1, 2, 128, 128).cuda(), torch.randn(1, 2 * 128, 128).cuda()
    with torch.no_grad(), split_linears(True):
        y = m(q, k, v)
    assert bool(torch.isfinite(y).all())
    md = CrossAttention(128, 4, 32, True).double()
    md.load_state_dict({k_: v_.double().cpu() for k_, v_ in m.state_dict().items()})
    with torch.no_grad():                      # float64 restatement of cvt_modules.py:95-173 with torch ops
        qd, kd, vd = q.double().cpu(), k.double().cpu(), v.double().cpu()
        qp, kp, vp = md.to_q(qd), md.to_k(kd), md.to_v(vd)
        b, n, Q, _ = qp.shape
        K = kp.shape[2]
        qh = qp.reshape(b, n, Q, 4, 32).permute(0, 3, 1, 2, 4)
        kh = kp.reshape(b, n, K, 4, 32).permute(0, 3, 1, 2, 4)
        dot = 32 ** -0.5 * torch.einsum("bmnqd,bmnkd->bmnqk", qh, kh)
        att = dot.permute(0, 1, 3, 2, 4).reshape(b, 4, Q, n * K).softmax(-1)
        a = torch.einsum("bmqk,bkmd->bqmd", att, vp.reshape(b, n * K, 4, 32)).reshape(b, Q, 128)
        z = md.prenorm(md.proj(a))
        z = md.postnorm(z + md.mlp(z))
    assert rel_max_err(y.cpu().double().reshape(b, Q, 128), z) < 1e-4
