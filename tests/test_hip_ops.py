"""GPU parity tests of the single operators exported by libhmvit (include/hmvit.h), each against
plain torch fp32 on the same seeded inputs.  Tolerances: f32 mode = fp32 round-off, f16 mode =
the 1e-3 relative budget of north_star split over the operators."""
import math
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_max_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import hmvit_amd  # noqa: F401
    from hmvit_amd import _lib
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_tr16_lane_mapping(lib):
    """ds_read_b64_tr_b16: lane l, element j must return lds[(l & 15) + 16 j + 64 (l >> 4)] when
    lane l points at element 4 l -- the V^T operand layout of the attention kernel relies on it."""
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    lib.check(lib.lib.hmvit_debug_tr16(out.data_ptr(), _stream()), "debug_tr16")
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    lane = np.arange(64)[:, None]
    want = (lane & 15) + 16 * np.arange(4)[None, :] + 64 * (lane >> 4)
    assert (got == want).all(), got


def test_layout_round_trip(lib):
    x = torch.randn(3, 64, 200, device="cuda")
    y = torch.empty(3, 200, 64, device="cuda")
    lib.check(lib.lib.hmvit_nchw_to_tokens(x.data_ptr(), y.data_ptr(), 3, 64, 200, _stream()), "nchw")
    assert torch.equal(y, x.permute(0, 2, 1))
    z = torch.empty_like(x)
    lib.check(lib.lib.hmvit_tokens_to_nchw(y.data_ptr(), z.data_ptr(), 3, 64, 200, _stream()), "tok")
    assert torch.equal(z, x)


@pytest.mark.parametrize("C", [64, 128, 256])
@pytest.mark.parametrize("prec", [0, 1])
def test_layernorm(lib, C, prec):
    torch.manual_seed(C + prec)
    n, P = 3, 333
    x = torch.randn(n, P, C, device="cuda") * 3 + 1
    gamma = torch.randn(2, C, device="cuda")
    beta = torch.randn(2, C, device="cuda")
    types = [1, 0, 1]
    y = torch.empty(n, P, C, device="cuda", dtype=torch.float32 if prec == 0 else torch.float16)
    lib.check(lib.lib.hmvit_layernorm(x.data_ptr(), y.data_ptr(), lib.i32_array(types), gamma.data_ptr(),
                                      beta.data_ptr(), n, P, C, prec, _stream()), "layernorm")
    ref = torch.stack([torch.nn.functional.layer_norm(x[i], (C,), gamma[t], beta[t], 1e-5)
                       for i, t in enumerate(types)])
    assert rel_max_err(y.float(), ref) < (2e-6 if prec == 0 else 6e-4)


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("M,N,K", [(200, 192, 64), (1000, 768, 256), (64, 64, 256), (129, 130, 128)])
@pytest.mark.parametrize("gelu,res,out_f32", [(0, 0, 0), (1, 0, 0), (0, 1, 1), (1, 1, 1)])
def test_linear(lib, prec, M, N, K, gelu, res, out_f32):
    torch.manual_seed(M + N + K)
    dt = torch.float32 if prec == 0 else torch.float16
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    y = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else dt)
    lib.check(lib.lib.hmvit_linear(a.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                   r.data_ptr() if res else None, y.data_ptr(), M, N, K, gelu, out_f32,
                                   prec, _stream()), "linear")
    ref = a.double() @ w.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.double()
    tol = 2e-6 if prec == 0 else (2e-6 if out_f32 else 6e-4)
    assert rel_max_err(y.double(), ref) < tol


@pytest.mark.parametrize("M,N,K", [(200, 192, 64), (1000, 768, 256), (129, 130, 128)])
@pytest.mark.parametrize("gelu,res", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_linear_split(lib, M, N, K, gelu, res):
    """HMVIT_PREC_SPLIT: f32 operands and result, products as (hi + lo) f16 halves - held to fp32 round-off class."""
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if res else None
    y = torch.empty(M, N, device="cuda")
    lib.check(lib.lib.hmvit_linear(a.data_ptr(), w.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, y.data_ptr(),
                                   M, N, K, gelu, 1, lib.PREC_SPLIT, _stream()), "linear")
    ref = a.double() @ w.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.double()
    assert rel_max_err(y.double(), ref) < 4e-6
    with pytest.raises(ValueError):          # f16 result is not offered in this mode
        lib.check(lib.lib.hmvit_linear(a.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), M, N, K, 0, 0, lib.PREC_SPLIT,
                                       _stream()), "linear")


def test_linear_rejects_bad_k(lib):
    a = torch.zeros(8, 48, device="cuda")
    with pytest.raises(ValueError):
        lib.check(lib.lib.hmvit_linear(a.data_ptr(), a.data_ptr(), None, None, a.data_ptr(), 8, 8, 48, 0, 1, 0,
                                       _stream()), "linear")


def test_warp_and_roi_match_golden(lib):
    """Sampling code of the attention gather vs the reference's warp_affine / ROI mask (g2)."""
    from oracle import hmvit_oracle as O
    g = load_golden("g2_warp.npz")
    src = g["src"]                                    # (1, C, H, W)
    _, C, H, W = src.shape
    cases = g["cases"].tolist()
    n = len(cases)
    T = torch.stack([O.rigid(*c).to(torch.float32) for c in cases]).cuda()
    ainv = torch.empty(n, 8, device="cuda")
    lib.check(lib.lib.hmvit_pair_affines(T.data_ptr(), ainv.data_ptr(), n, H, W, 0.4, 4.0, _stream()), "aff")
    tok = src[0].permute(1, 2, 0).reshape(1, H * W, C).repeat(n, 1, 1).contiguous().cuda()
    dst = torch.empty_like(tok)
    roi = torch.empty(n, H * W, device="cuda")
    lib.check(lib.lib.hmvit_warp_affine(tok.data_ptr(), ainv.data_ptr(), dst.data_ptr(), roi.data_ptr(), n, H,
                                        W, C, _stream()), "warp")
    got = dst.reshape(n, H, W, C).permute(0, 3, 1, 2).cpu()
    assert float((got - g["bilinear"]).abs().max()) < 5e-5
    assert int((roi.reshape(n, H, W).cpu() != g["roi"]).sum()) == 0
    assert float(ainv[0, 6]) == 1.0 and float(ainv[3, 6]) == 0.0   # identity flag


@pytest.mark.parametrize("precision", ["f32", "f16", "split"])
@pytest.mark.parametrize("C,window,partition", [(64, 4, 0), (64, 4, 1), (256, 8, 0), (256, 8, 1), (128, 8, 1)])
def test_window_attention_operator(lib, precision, C, window, partition):
    """hmvit_window_attention on its own (the exported operator, not the fused forward): projected Q / K' / V' maps of three
    agents of mixed types in, attention output of every ego out, against torch fp64 composed from the oracle's warp, ROI mask
    and window partition (which g1 / g2 pin to the reference): warp the source's K' / V' into the ego's frame, add the
    projection biases, position bias, -inf on masked keys, one softmax over all agents' keys of the window."""
    from hmvit_amd import weights
    from oracle import hmvit_oracle as O
    L, H, W, E = 3, 16, 24, 2
    M, d, n = C // 32, 32, window * window
    modes = [1, 0, 1]
    cav = [1, 1, 1]
    g = torch.Generator().manual_seed(7 + C + window + partition)
    rn = lambda *s: torch.randn(*s, generator=g)
    q, kv = rn(L, H * W, C) * 0.4, rn(L, E, 2, H * W, C) * 0.4
    b_q, b_kv = rn(2, C) * 0.1, rn(2, 2, 2 * C) * 0.1
    table = rn((2 * window - 1) ** 2, M)
    poses = [O.rigid(0.0, 0.0, 0.0), O.rigid(0.25, 3.0, -2.0), O.rigid(-0.4, -2.5, 1.5)]
    pw = O.pairwise_from_poses(poses, L)[None].float()                         # (1, L, L, 4, 4)
    ego_e = [0 if modes[i] == modes[0] else 1 for i in range(L)]              # variant index = order of first appearance
    variant_type = [modes[0], 1 - modes[0]]

    # ---- reference (fp64) ----
    idx = O.relative_position_index(window)
    bias = table.double()[idx].permute(2, 0, 1)                               # (M, n, n)
    ref = torch.zeros(L, H * W, C, dtype=torch.float64)
    grid = partition == 1
    for i in range(L):
        t_to_i = pw[:, :, i]                                                  # (1, L, 4, 4): source j -> ego i
        e = ego_e[i]
        kmap = kv[:, e, 0].reshape(L, H, W, C).permute(0, 3, 1, 2)[None]      # (1, L, C, H, W)
        vmap = kv[:, e, 1].reshape(L, H, W, C).permute(0, 3, 1, 2)[None]
        kw = O.warp_agents(kmap, t_to_i, 0.4, 2.0).double() + b_kv[modes[i], modes, :C].double()[None, :, :, None, None]
        vw = O.warp_agents(vmap, t_to_i, 0.4, 2.0).double() + b_kv[modes[i], modes, C:].double()[None, :, :, None, None]
        vis = O.roi_and_cav_mask(H, W, torch.tensor([cav], dtype=torch.float32), t_to_i, 0.4, 2.0)   # (1, H, W, 1, L)
        qi = (q[i].double() + b_q[modes[i]].double()).reshape(1, 1, H, W, C).permute(0, 1, 4, 2, 3)
        qp = O._partition(qi, window, grid)[0, 0].reshape(-1, n, M, d)                                   # (nW, n, M, d)
        kp = O._partition(kw, window, grid)[0].reshape(L, -1, n, M, d)
        vp = O._partition(vw, window, grid)[0].reshape(L, -1, n, M, d)
        mp = O._partition(vis.permute(0, 4, 3, 1, 2).double(), window, grid)[0].reshape(L, -1, n)        # (L, nW, n)
        sim = torch.einsum("wqhd,lwkhd->whqlk", qp, kp) + bias[None, :, :, None, :]
        sim = sim.masked_fill(mp.permute(1, 0, 2)[:, None, None] == 0, float("-inf"))
        att = torch.softmax(sim.reshape(*sim.shape[:3], -1), dim=-1).reshape(sim.shape)
        o = torch.einsum("whqlk,lwkhd->wqhd", att, vp)
        X, Y = H // window, W // window
        ref[i] = O._unpartition(o.reshape(1, 1, X, Y, window, window, C), grid)[0, 0].permute(1, 2, 0).reshape(H * W, C)

    # ---- operator ----
    prec = {"f32": lib.PREC_F32, "f16": lib.PREC_F16, "split": lib.PREC_SPLIT}[precision]
    s = 1.4426950408889634 if precision == "f16" else 1.0                    # the f16 kernels take logits in log2 units
    dt = torch.float16 if precision == "f16" else torch.float32
    dq, dkv = (q * s).to(dt).cuda(), kv.to(dt).cuda()
    frag = (weights.bias_fragments(table, window) * s).cuda()
    dbq, dbkv = (b_q * s).cuda(), b_kv.cuda()
    ainv = torch.empty(L * L, 8, device="cuda")
    lib.check(lib.lib.hmvit_pair_affines(pw.cuda().contiguous().data_ptr(), ainv.data_ptr(), L * L, H, W, 0.4, 2.0, _stream()), "aff")
    out = torch.zeros(L, H * W, C, device="cuda", dtype=dt)
    lib.check(lib.lib.hmvit_window_attention(dq.data_ptr(), dkv.data_ptr(), dbq.data_ptr(), dbkv.data_ptr(), frag.data_ptr(),
                                             ainv.data_ptr(), lib.i32_array(modes), lib.i32_array(cav), lib.i32_array(ego_e),
                                             out.data_ptr(), 1, L, L, L, E, C, H, W, window, partition, prec, 0, _stream()),
              "window_attention")
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < {"f32": 2e-5, "f16": 3e-3, "split": 2e-5}[precision], err


@pytest.mark.parametrize("M,n_mat,bias,res,a_scale,w_scale", [
    (1000, 1, True, False, 1.0, 1.0), (129, 1, True, True, 1.0, 1.0), (4096, 3, True, False, 1.0, 1.0), (300, 2, False, False, 1.0, 1.0),
    (640, 1, True, True, 1e-7, 1.0), (640, 1, True, False, 3e5, 1e-4), (640, 3, False, False, 1.0, 300.0)])
def test_linear16_matches_float64(lib, M, n_mat, bias, res, a_scale, w_scale):
    """hmvit_linear16 (the training path's skinny Linear: x16 tiles, split-f16 products, per-token / per-matrix power-of-two
    scaling) against torch.nn.functional.linear in float64: fp32 round-off class whatever the magnitudes (1e-7 rows are
    back-propagated gradients, 3e5 un-normalised residual streams), ragged M, up to three matrices over one pass of the rows."""
    torch.manual_seed(M + n_mat)
    a = (torch.randn(M, 256, device="cuda") * a_scale)
    a[::7] *= 50.0                                              # rows of very different magnitude next to each other
    w = torch.randn(256 * n_mat, 256, device="cuda") * w_scale / 16
    b = torch.randn(256 * n_mat, device="cuda") * a_scale * w_scale if bias else None
    r = torch.randn(M, 256, device="cuda") * a_scale * w_scale if res else None
    y = torch.full((M, 256 * n_mat), float("nan"), device="cuda")
    ws = torch.empty(65536 * n_mat + 64, device="cuda")
    lib.check(lib.lib.hmvit_linear16(a.data_ptr(), w.data_ptr(), b.data_ptr() if bias else None, r.data_ptr() if res else None,
                                     y.data_ptr(), M, n_mat, ws.data_ptr(), _stream()), "linear16")
    ref = torch.nn.functional.linear(a.double(), w.double(), b.double() if bias else None)
    if res:
        ref = ref + r.double()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    row_err = ((y.double() - ref).abs().amax(1) / ref.abs().amax(1)).max().item()    # every row against its own magnitude
    print(f"\nlinear16 M={M} n_mat={n_mat} a_scale={a_scale:g} w_scale={w_scale:g}: rel-max {err:.2e}, worst row {row_err:.2e}")
    assert torch.isfinite(y).all()
    assert err < 2e-6 and row_err < 5e-6
    if res:                                                      # the residual may alias the output (accumulation in place)
        y2 = r.clone()
        lib.check(lib.lib.hmvit_linear16(a.data_ptr(), w.data_ptr(), b.data_ptr() if bias else None, y2.data_ptr(), y2.data_ptr(),
                                         M, n_mat, ws.data_ptr(), _stream()), "linear16")
        assert torch.equal(y2, y)


@pytest.mark.gpu
@pytest.mark.parametrize("dtypes", [(torch.float64, torch.int64, torch.float32), (torch.int32, torch.int32, torch.bool),
                                    (torch.float16, torch.int64, torch.uint8), (torch.float32, torch.float32, torch.float64)])
@pytest.mark.parametrize("pw_dtype", [None, torch.float32, torch.float64])
def test_small_inputs_read_back_in_one_launch(dtypes, pw_dtype):
    """hmvit_pack_small (one launch, one copy) returns what the aten formulation of HeteroFusion._host_small returns: mode / record_len /
    mask in every dtype a caller may hold them in, and the "all self transforms are the identity" flag - exact comparison, one
    perturbed diagonal block turns it off."""
    import hmvit_amd
    from hmvit_amd.fusion import _FusionBase
    B, L = 3, 5
    g = torch.Generator().manual_seed(7)
    mode = torch.randint(0, 2, (B, L), generator=g)
    rl = torch.randint(1, L + 1, (B,), generator=g)
    mask = torch.randint(0, 2, (B, L), generator=g)
    host = [mode.to(dtypes[0]), rl.to(dtypes[1]), mask.to(dtypes[2])]
    dev = [t.cuda() for t in host]
    for perturb in (False, True):
        pw = None
        if pw_dtype is not None:
            pw = torch.eye(4, dtype=pw_dtype).repeat(B, L, L, 1, 1)
            pw[:, 0, 1, 0, 3] = 2.5                        # off-diagonal pairs may be anything
            if perturb:
                pw[2, 3, 3, 1, 3] += 1e-6 if pw_dtype == torch.float32 else 1e-13
        want = _FusionBase._host_small(*host, pw)                                   # CPU tensors: the aten formulation
        got = _FusionBase._pack_small_on_device(dev, pw.cuda() if pw is not None else None)
        assert got is not None and got == want
        assert _FusionBase._host_small(*dev, pw.cuda() if pw is not None else None) == want
        if pw is not None:
            assert want[3] == (2 | (0 if perturb else 1))       # bit 0: identity self transforms, bit 1: rigid pairs (still true here)
    if pw_dtype is not None:                                # a sheared / scaled pair turns the rigidity bit off, a rotation keeps it
        pw = torch.eye(4, dtype=pw_dtype).repeat(B, L, L, 1, 1)
        c, s_ = math.cos(0.7), math.sin(0.7)
        pw[1, 2, 4, :2, :2] = torch.tensor([[c, -s_], [s_, c]], dtype=pw_dtype)
        assert _FusionBase._pack_small_on_device(dev, pw.cuda())[3] == 3 == _FusionBase._host_small(*host, pw)[3]
        pw[1, 2, 4, 0, 0] = c * 1.03
        assert _FusionBase._pack_small_on_device(dev, pw.cuda())[3] == 1 == _FusionBase._host_small(*host, pw)[3]
        pw[1, 2, 4, 0, 0] = float("nan")
        assert _FusionBase._pack_small_on_device(dev, pw.cuda())[3] == 1 == _FusionBase._host_small(*host, pw)[3]
    # mixed placement / an unsupported dtype: no device path, the aten formulation takes over
    assert _FusionBase._pack_small_on_device([dev[0], host[1], dev[2]], None) is None
    assert _FusionBase._pack_small_on_device([dev[0].to(torch.int16), dev[1], dev[2]], None) is None
    assert _FusionBase._host_small(dev[0].to(torch.int16), dev[1], dev[2]) == _FusionBase._host_small(*host)
