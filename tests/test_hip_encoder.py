"""GPU parity tests of the LiDAR BEV encoder (csrc/enc.hip, hm-vit_amd/pointpillar.py) against the
golden vector frozen from the reference's PointPillar (g7) and against torch fp64 convolutions."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_max_err
from oracle import pointpillar_oracle as PO

pytestmark = pytest.mark.gpu
TOL = {"f32": 1e-4, "split": 1e-4, "f16": 2e-3}   # 20+ chained f16 convolutions: looser than the fusion's 1e-3


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _golden_inputs():
    g = load_golden("g7_pointpillar.npz")
    nx, ny = [int(v) for v in g["grid"]]
    args = PO.make_args(nx, ny)
    sd = PO.random_state_dict(args, g["seed_weights"])
    pillars = PO.synthetic_pillars(int(g["n_agents"]), int(g["n_per_agent"]), nx, ny, args, int(g["seed_pillars"]))
    return g, args, sd, pillars


def test_pfn_matches_golden():
    from hmvit_amd import _lib
    g, args, sd, (vf, vc, vn) = _golden_inputs()
    p = "pillar_vfe.pfn_layers.0"
    scale = sd[f"{p}.norm.weight"] / torch.sqrt(sd[f"{p}.norm.running_var"] + 1e-3)
    w = (sd[f"{p}.linear.weight"] * scale[:, None]).contiguous().cuda()
    shift = (sd[f"{p}.norm.bias"] - sd[f"{p}.norm.running_mean"] * scale).contiguous().cuda()
    vf, vc, vn = vf.cuda(), vc.cuda(), vn.cuda()
    nx, ny = [int(v) for v in g["grid"]]
    out = torch.empty(vf.shape[0], 64, device="cuda")
    canvas = torch.zeros(2, ny, nx, 64, device="cuda")
    vs = (ctypes.c_float * 3)(*args["voxel_size"])
    rng = (ctypes.c_float * 6)(*args["lidar_range"])
    _lib.check(_lib.lib.hmvit_pfn_scatter(vf.data_ptr(), vc.data_ptr(), vn.data_ptr(), w.data_ptr(), shift.data_ptr(),
                                          canvas.data_ptr(), out.data_ptr(), vf.shape[0], nx, ny, 2, None, vs, rng, 0, _stream()),
               "pfn")
    assert rel_max_err(out.cpu(), g["pillar_features"]) < 1e-5
    ref = PO.scatter(g["pillar_features"], vc.cpu(), 2, ny, nx)                 # (2, 64, ny, nx)
    assert rel_max_err(canvas.permute(0, 3, 1, 2).cpu(), ref) < 1e-5


def test_pfn_scatter_drops_out_of_range_pillars():
    """an agent index >= n_agents or a coordinate outside the grid must not write out of bounds; such pillars
    are dropped and counted (the reference's indexed scatter raises, point_pillar_scatter.py:30-40)."""
    from hmvit_amd import _lib
    g, args, sd, (vf, vc, vn) = _golden_inputs()
    nx, ny = [int(v) for v in g["grid"]]
    vc = vc.clone()
    bad = [3, 17, 101, 250]
    vc[bad[0], 0] = 2          # agent index beyond the canvas planes
    vc[bad[1], 2] = ny         # y outside
    vc[bad[2], 3] = -1         # x outside
    vc[bad[3], 0] = 1 << 20
    w = torch.randn(64, 10).cuda()
    shift = torch.randn(64).cuda()
    # guard planes around the canvas: any out-of-bounds write of a near miss lands in them
    buf = torch.zeros(4, ny, nx, 64, device="cuda")
    canvas = buf[1:3]
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    vs = (ctypes.c_float * 3)(*args["voxel_size"])
    rng = (ctypes.c_float * 6)(*args["lidar_range"])
    vf, vcd, vn = vf.cuda(), vc.cuda(), vn.cuda()
    _lib.check(_lib.lib.hmvit_pfn_scatter(vf.data_ptr(), vcd.data_ptr(), vn.data_ptr(), w.data_ptr(), shift.data_ptr(),
                                          canvas.data_ptr(), None, vf.shape[0], nx, ny, 2, cnt.data_ptr(), vs, rng, 0,
                                          _stream()), "pfn")
    torch.cuda.synchronize()
    assert int(cnt.item()) == len(bad)
    assert float(buf[0].abs().max()) == 0.0 and float(buf[3].abs().max()) == 0.0
    good = torch.ones(vc.shape[0], dtype=torch.bool)
    good[bad] = False
    occupied = torch.zeros(2, ny, nx, dtype=torch.bool)
    occupied[vc[good, 0].long(), vc[good, 2].long(), vc[good, 3].long()] = True
    assert bool(((canvas.abs().sum(-1) > 0).cpu() <= occupied).all())


@pytest.mark.parametrize("drop", [0, 3])
def test_pfn_scatter_reports_the_canvas_range(drop):
    """hmvit_conv_range(NULL, 0, slot) before the scatter: the slot receives max |canvas| exactly (the value the first split
    convolution would otherwise measure with a pass over the canvas) - also when the pillar count is not a multiple of the four
    a workgroup takes (per-wavefront path of the last workgroup), and dropped pillars do not count."""
    from hmvit_amd import _lib
    g, args, sd, (vf, vc, vn) = _golden_inputs()
    nx, ny = [int(v) for v in g["grid"]]
    n = vf.shape[0] - drop
    vf, vc, vn = vf[:n].clone(), vc[:n].clone(), vn[:n].clone()
    torch.manual_seed(3)
    w, shift = torch.randn(64, 10).cuda(), torch.randn(64).cuda()
    vf[5] *= 40.0                     # the largest value sits in one pillar ...
    vf[9] *= 90.0
    vc[9, 0] = 7                      # ... and an even larger one in a pillar that is dropped (agent index out of range)
    canvas = torch.zeros(2, ny, nx, 64, device="cuda")
    vs = (ctypes.c_float * 3)(*args["voxel_size"])
    rng = (ctypes.c_float * 6)(*args["lidar_range"])
    vf, vc, vn = vf.cuda(), vc.cuda(), vn.cuda()
    _lib.announce_output_range(canvas)
    slot = _lib.range_of(canvas)
    _lib.check(_lib.lib.hmvit_pfn_scatter(vf.data_ptr(), vc.data_ptr(), vn.data_ptr(), w.data_ptr(), shift.data_ptr(),
                                          canvas.data_ptr(), None, n, nx, ny, 2, None, vs, rng, _lib.PREC_SPLIT, _stream()), "pfn")
    torch.cuda.synchronize()
    got = slot[:1].view(torch.float32)
    assert float(got) == float(canvas.abs().max()) > 0.0
    # the slot was consumed: a second call without an announcement leaves it alone
    slot.zero_()
    _lib.check(_lib.lib.hmvit_pfn_scatter(vf.data_ptr(), vc.data_ptr(), vn.data_ptr(), w.data_ptr(), shift.data_ptr(),
                                          canvas.data_ptr(), None, n, nx, ny, 2, None, vs, rng, _lib.PREC_SPLIT, _stream()), "pfn")
    torch.cuda.synchronize()
    assert int(slot[0]) == 0


def test_pointpillar_rejects_wrong_voxel_layout():
    import hmvit_amd
    args = PO.make_args(64, 48)
    net = hmvit_amd.PointPillar(args, precision="f32").cuda().eval()
    net.set_return_features()
    batch = {"processed_lidar": {"voxel_features": torch.zeros(10, 16, 4).cuda(), "voxel_coords": torch.zeros(10, 4, dtype=torch.int32).cuda(),
                                 "voxel_num_points": torch.ones(10, dtype=torch.int32).cuda()}}
    with pytest.raises(ValueError):
        net(batch)


@pytest.mark.parametrize("prec", [0, 1, 2])                      # HMVIT_PREC_F32 / F16 / SPLIT (f32 maps, split-f16 products)
@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W", [(64, 64, 3, 2, 1, 20, 28), (128, 256, 3, 1, 1, 9, 7),
                                                       (384, 256, 3, 2, 1, 10, 12), (256, 14, 1, 1, 0, 6, 5)])
def test_conv2d(prec, cin, cout, k, stride, pad, H, W):
    from hmvit_amd import _lib
    torch.manual_seed(cin + cout)
    dt = torch.float16 if prec == 1 else torch.float32
    x = torch.randn(2, cin, H, W, device="cuda").to(dt)
    w = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).to(dt)
    b = torch.randn(cout, device="cuda")
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride, pad))
    Ho, Wo = ref.shape[-2:]
    y = torch.empty(2, Ho, Wo, cout + 8, device="cuda", dtype=dt).fill_(7)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wn = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    _lib.check(_lib.lib.hmvit_conv2d(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), y.data_ptr(), 2, H, W, cin, cout, k,
                                     stride, pad, 1, cout + 8, 3, 0, 0, prec, _stream()), "conv")
    got = y[..., 3:3 + cout].permute(0, 3, 1, 2).double()
    assert rel_max_err(got, ref) < {0: 2e-6, 1: 1.5e-3, 2: 4e-6}[prec]
    assert bool((y[..., :3] == 7).all()) and bool((y[..., 3 + cout:] == 7).all())   # channel window respected


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("s", [1, 2, 4])
def test_deconv2d(prec, s):
    from hmvit_amd import _lib
    torch.manual_seed(s)
    dt = torch.float16 if prec == 1 else torch.float32
    cin, cout, H, W = 128, 128, 5, 6
    x = torch.randn(2, cin, H, W, device="cuda").to(dt)
    w = (torch.randn(cin, cout, s, s, device="cuda") / cin ** 0.5).to(dt)
    b = torch.randn(cout, device="cuda")
    ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), s))
    y = torch.empty(2, H * s, W * s, cout, device="cuda", dtype=dt)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wn = w.permute(2, 3, 1, 0).reshape(s * s * cout, cin).contiguous()
    _lib.check(_lib.lib.hmvit_conv2d(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), y.data_ptr(), 2, H, W, cin, cout, 1, 1,
                                     0, 1, cout, 0, s, 0, prec, _stream()), "deconv")
    assert rel_max_err(y.permute(0, 3, 1, 2).double(), ref) < {0: 2e-6, 1: 1.5e-3, 2: 4e-6}[prec]


@pytest.mark.parametrize("precision", ["f32", "f16", "split"])
def test_pointpillar_encoder_matches_golden(precision):
    import hmvit_amd
    g, args, sd, (vf, vc, vn) = _golden_inputs()
    net = hmvit_amd.PointPillar(args, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    net.set_return_features()
    batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}}
    y = net(batch).cpu()
    assert y.shape == g["out"].shape
    assert rel_max_err(y, g["out"]) < TOL[precision]


def test_pointpillar_heads_and_errors():
    import hmvit_amd
    g, args, sd, (vf, vc, vn) = _golden_inputs()
    net = hmvit_amd.PointPillar(args, precision="f32")
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}}
    out = net(batch)
    feats = PO.point_pillar_features(vf, vc, vn, sd, args, 2)
    psm = F.conv2d(feats, sd["cls_head.weight"], sd["cls_head.bias"])
    rm = F.conv2d(feats, sd["reg_head.weight"], sd["reg_head.bias"])
    assert rel_max_err(out["psm"].cpu(), psm) < 1e-4 and rel_max_err(out["rm"].cpu(), rm) < 1e-4
    with pytest.raises(RuntimeError):
        net.train()(batch)


@pytest.mark.parametrize("precision", ["f32", "f16", "split"])
def test_hetero_decoder_matches_golden(precision):
    import numpy as np
    import hmvit_amd
    from oracle import decoder_oracle as DO
    g = load_golden("g8_decoder.npz")
    params = DO.make_params()
    net = hmvit_amd.HeteroDecoder(params, precision=precision)
    net.load_state_dict(DO.random_state_dict(params, g["seed_weights"]), strict=True)
    net = net.cuda().eval()
    x = torch.from_numpy(np.random.RandomState(int(g["seed_x"])).standard_normal((3, 1, 256, 12, 10)).astype(np.float32))
    psm, rm = net(x.cuda(), g["mode"].cuda(), use_upsample=False)
    assert rel_max_err(psm.cpu(), g["psm"]) < TOL[precision]
    assert rel_max_err(rm.cpu(), g["rm"]) < TOL[precision]
    with pytest.raises(NotImplementedError):
        net(x.cuda(), g["mode"].cuda())


@pytest.mark.parametrize("seed", range(4))
def test_pointpillar_random_sweep_vs_oracle(seed):
    """Random grid sizes, agent counts and pillar counts (incl. an agent without pillars): encoder features against the
    oracle in both precisions."""
    import numpy as np
    import hmvit_amd
    rs = np.random.RandomState(500 + seed)
    nx, ny = int(rs.choice([32, 64, 96])), int(rs.choice([32, 48]))
    args = PO.make_args(nx, ny)
    sd = PO.random_state_dict(args, seed=600 + seed)
    n_agents = int(rs.choice([1, 2, 3]))
    vf, vc, vn = PO.synthetic_pillars(n_agents, int(rs.choice([20, 150, 400])), nx, ny, args, seed=700 + seed)
    if n_agents == 3:                                   # drop every pillar of the middle agent
        keep = vc[:, 0] != 1
        vf, vc, vn = vf[keep], vc[keep], vn[keep]
    ref = PO.point_pillar_features(vf, vc, vn, sd, args, n_agents)
    batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()},
             "record_len": torch.tensor([n_agents])}
    for precision in ("f32", "f16"):
        net = hmvit_amd.PointPillar(args, precision=precision)
        net.load_state_dict(sd, strict=True)
        net = net.cuda().eval()
        net.set_return_features()
        y = net(batch).cpu()
        assert y.shape == ref.shape
        assert rel_max_err(y, ref) < TOL[precision], (precision, nx, ny, n_agents)


@pytest.mark.parametrize("N,cin,cout,H,W,res,out_f32", [(5, 128, 256, 44, 75, True, False), (8, 64, 64, 64, 66, False, False),
                                                        (6, 192, 136, 40, 48, False, True), (6, 192, 136, 40, 48, True, False)])
def test_conv3x3_patch_kernel(N, cin, cout, H, W, res, out_f32):
    """The 3 x 3 / stride 1 / pad 1 f16 path (k_conv3: input patch staged in LDS once per channel slab) at sizes with ragged
    tiles (H, W not multiples of 8 / 16, Cout not a multiple of the channel tile), with and without the residual operand and
    the f32 output, against torch on the same f16 operands; and identical to the generic kernel's result."""
    import os
    from hmvit_amd import _lib
    torch.manual_seed(N * cin + cout)
    x = torch.randn(N, cin, H, W, device="cuda").half()
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).half()
    b = torch.randn(cout, device="cuda")
    r = torch.randn(N, cout, H, W, device="cuda").half() if res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    if res:
        ref = ref + r.double()
    ref = F.relu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wn = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    rn = r.permute(0, 2, 3, 1).contiguous() if res else None

    def run(generic=False):
        y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.float32 if out_f32 else torch.float16)
        _lib.check(_lib.lib.hmvit_conv2d_ex(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), rn.data_ptr() if res else None, y.data_ptr(), N, H, W,
                                            cin, cout, 3, 1, 1, 1, 2 if generic else 0, 1 if out_f32 else 0, _lib.PREC_F16, _stream()),
                   "conv2d_ex")
        return y
    y = run()
    assert rel_max_err(y.permute(0, 3, 1, 2).double(), ref) < (2e-4 if out_f32 else 1.5e-3)
    y_generic = run(generic=True)
    assert rel_max_err(y.double(), y_generic.double()) < 1e-3      # same products, different summation order per tap


@pytest.mark.parametrize("N,cin,cout,H,W,res,up2", [(5, 128, 256, 44, 75, True, False), (8, 64, 64, 64, 66, False, False),
                                                    (6, 96, 136, 40, 48, False, False), (6, 128, 128, 42, 80, False, True)])
def test_conv3x3_patch_kernel_split(N, cin, cout, H, W, res, up2):
    """The same patch-in-LDS kernel in split precision (f32 map / weights / residual / result, split-f16 products): ragged tiles, a
    channel count that is a multiple of 32 but not of 64, the residual operand and the upsampled-input mode, against torch fp64
    at fp32 round-off class; and against the generic split kernel."""
    from hmvit_amd import _lib
    torch.manual_seed(N * cin + cout + 1)
    h, w_ = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(N, cin, h, w_, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    b = torch.randn(cout, device="cuda")
    r = torch.randn(N, cout, H, W, device="cuda") if res else None
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
    ref = F.conv2d(xin, w.double(), b.double(), 1, 1)
    if res:
        ref = ref + r.double()
    ref = F.relu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous()
    wn = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    rn = r.permute(0, 2, 3, 1).contiguous() if res else None

    def run(generic=False):
        y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.float32)
        _lib.check(_lib.lib.hmvit_conv2d_ex(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), rn.data_ptr() if res else None, y.data_ptr(), N, H, W,
                                            cin, cout, 3, 1, 1, 1, (2 if generic else 0) | (1 if up2 else 0), 1, _lib.PREC_SPLIT, _stream()),
                   "conv2d_ex")
        return y
    y = run()
    assert rel_max_err(y.permute(0, 3, 1, 2).double(), ref) < 4e-6
    y_generic = run(generic=True)
    assert rel_max_err(y.double(), y_generic.double()) < 4e-6


def test_conv3x3_patch_kernel_upsampled_input():
    """Same kernel with the decoder's operand mode: the input is the nearest x2 upsampling of a half-size map."""
    import os
    from hmvit_amd import _lib
    torch.manual_seed(5)
    N, cin, cout, h, w_ = 6, 128, 128, 21, 40                     # logical (upsampled) size 42 x 80
    x = torch.randn(N, cin, h, w_, device="cuda").half()
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).half()
    b = torch.randn(cout, device="cuda")
    ref = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), 1, 1))
    xn = x.permute(0, 2, 3, 1).contiguous()
    wn = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()

    def run(generic=False):
        y = torch.empty(N, 2 * h, 2 * w_, cout, device="cuda", dtype=torch.float16)
        _lib.check(_lib.lib.hmvit_conv2d_ex(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, 2 * h, 2 * w_, cin, cout, 3, 1, 1,
                                            1, 3 if generic else 1, 0, _lib.PREC_F16, _stream()), "conv2d_ex")
        return y
    y = run()
    assert rel_max_err(y.permute(0, 3, 1, 2).double(), ref) < 1.5e-3
    y_generic = run(generic=True)
    assert rel_max_err(y.double(), y_generic.double()) < 1e-3


@pytest.mark.parametrize("precision", ["f32", "f16", "split"])
def test_naive_compressor_matches_golden(precision):
    """NaiveCompressor (naive_compress.py:5-28) against the reference's own forward (g15) and the model's `compression` option."""
    import numpy as np
    import hmvit_amd
    from oracle import decoder_oracle as DO
    g = load_golden("g15_compressor.npz")
    net = hmvit_amd.NaiveCompressor(256, 4, precision=precision)
    net.load_state_dict(DO.compressor_state_dict(256, 4, seed=g["seed_weights"]), strict=True)
    net = net.cuda().eval()
    x = torch.from_numpy(np.random.RandomState(int(g["seed_x"])).standard_normal((3, 256, 10, 12)).astype(np.float32))
    y = net(x.cuda()).cpu()
    assert rel_max_err(y, g["out"]) < TOL[precision]


@pytest.mark.parametrize("prec_name", ["split", "f16"])
@pytest.mark.parametrize("N,cin,cout,H,W,res,up2", [(5, 128, 256, 44, 75, True, False), (8, 64, 64, 64, 66, False, False),
                                                    (6, 192, 136, 40, 48, False, False), (6, 128, 128, 42, 80, False, True),
                                                    (5, 256, 256, 64, 64, False, False), (3, 384, 256, 48, 64, False, False)])
def test_conv3x3_ring_kernel_is_bit_identical_to_the_patch_kernel(prec_name, N, cin, cout, H, W, res, up2):
    """The LDS-DMA ring kernel (k_conv3r: weight slabs copied global -> LDS from the prepared image, hmvit_conv3x3_image) against
    the register-staged patch kernel on the same operands: the MFMA order is the same, so every output bit must be; plus the
    float64 convolution at the precision's bound.  Shapes: ragged tiles, 64- and 128-channel tiles with a ragged last one, one /
    many channel slabs (the patch replacement path), residual operand, upsampled input, the PointPillar block-3 shape."""
    from hmvit_amd import _lib
    prec = _lib.PREC_SPLIT if prec_name == "split" else _lib.PREC_F16
    dt = torch.float32 if prec_name == "split" else torch.float16
    torch.manual_seed(N * cin + cout + 7)
    h, w_ = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(N, cin, h, w_, device="cuda").to(dt)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).to(dt)
    b = torch.randn(cout, device="cuda")
    r = torch.randn(N, cout, H, W, device="cuda").to(dt) if res else None
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
    ref = F.conv2d(xin, w.double(), b.double(), 1, 1)
    if res:
        ref = ref + r.double()
    ref = F.relu(ref)
    xn = x.permute(0, 2, 3, 1).contiguous()
    rn = r.permute(0, 2, 3, 1).contiguous() if res else None
    wmax = 0.0
    wk = w.float()
    if prec_name == "split":
        wk, wmax = _lib.prescale_weights(wk)
    wn = wk.permute(0, 2, 3, 1).reshape(cout, -1).to(dt).contiguous()
    img = _lib.conv3_image(wn, cout, cin, 3, 1, 1, prec, wmax)
    assert img is not None and img.numel() == _lib.lib.hmvit_conv3x3_image_bytes(cout, cin, prec)

    def run(ring):
        y = torch.empty(N, H, W, cout, device="cuda", dtype=dt)
        if prec_name == "split":
            _lib.conv_range(xn, wmax, y, _stream())
        if ring:
            _lib.use_conv_image(img)
        _lib.check(_lib.lib.hmvit_conv2d_ex(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), rn.data_ptr() if res else None, y.data_ptr(), N, H, W,
                                            cin, cout, 3, 1, 1, 1, 1 if up2 else 0, 0, prec, _stream()), "conv2d_ex")
        return y
    y_ring, y_patch = run(True), run(False)
    # below 256 workgroups the library keeps the generic kernel for calls WITHOUT an image (the ring kernel is taken from 128
    # on): other summation order there, so round-off-level agreement instead of equal bits
    n_wg = N * -(-H // 8) * -(-W // 16) * -(-cout // (64 if cout <= 64 else 128))
    if n_wg >= 256:
        assert torch.equal(y_ring, y_patch), float((y_ring.double() - y_patch.double()).abs().max())
    else:
        assert n_wg >= 128 and rel_max_err(y_ring.double(), y_patch.double()) < (4e-6 if prec_name == "split" else 1e-3)
    assert rel_max_err(y_ring.permute(0, 3, 1, 2).double(), ref) < (4e-6 if prec_name == "split" else 1.5e-3)
    # the image is consumed by ONE call: the next call without it must take the patch kernel again (same bits either way)
    assert torch.equal(run(False), y_patch)


def test_conv3x3_image_is_declined_where_the_ring_kernel_does_not_apply():
    from hmvit_amd import _lib
    w = torch.randn(64, 9 * 64, device="cuda")
    assert _lib.conv3_image(w, 64, 64, 3, 2, 1, _lib.PREC_SPLIT, -2.0) is None        # stride 2
    assert _lib.conv3_image(w, 64, 64, 1, 1, 0, _lib.PREC_SPLIT, -2.0) is None        # 1 x 1
    assert _lib.conv3_image(w, 64, 64, 3, 1, 1, _lib.PREC_F32, 0.0) is None           # exact-f32 mode
    assert _lib.conv3_image(w, 64, 64, 3, 1, 1, _lib.PREC_SPLIT, 3.0) is None         # split weights not pre-scaled
    assert _lib.lib.hmvit_conv3x3_image_bytes(64, 48, _lib.PREC_SPLIT) == 0           # Cin not a multiple of the slab depth
    assert _lib.lib.hmvit_conv3x3_image_bytes(64, 96, _lib.PREC_F16) == 0


@pytest.mark.parametrize("N,cin,cout,k,stride,pad,H,W,deconv", [(5, 384, 256, 3, 2, 0, 64, 64, 0),     # strided, no padding
                                                              (4, 64, 64, 3, 3, 1, 50, 38, 0),       # narrow channel tile, ragged pixels, stride 3
                                                              (3, 128, 256, 1, 1, 0, 20, 12, 0),     # 1 x 1, small map (64-pixel tiles)
                                                              (2, 256, 128, 1, 1, 0, 24, 20, 2),     # ConvTranspose2d(kernel = stride = 2)
                                                              (2, 64, 136, 3, 1, 0, 21, 19, 0)])     # ragged channel tile, pad 0
def test_conv_gemm_ring_is_bit_identical_to_the_register_staged_kernel(N, cin, cout, k, stride, pad, H, W, deconv):
    """Generic implicit-GEMM kernel in split mode with its weight slabs on the LDS-DMA ring (hmvit_conv_gemm_image, kind 1) against
    the same kernel staging the weights through registers: same MFMA order, so equal bits; plus the float64 convolution."""
    from hmvit_amd import _lib
    torch.manual_seed(N * cin + cout + k)
    x = torch.randn(N, cin, H, W, device="cuda")
    b = torch.randn(cout, device="cuda")
    if deconv:
        w = torch.randn(cin, cout, deconv, deconv, device="cuda") / cin ** 0.5
        ref = F.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), deconv))
        wk, wmax = _lib.prescale_weights(w)
        wn = wk.permute(2, 3, 1, 0).reshape(deconv * deconv * cout, cin).contiguous()
        Ho, Wo = H * deconv, W * deconv
    else:
        w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride, pad))
        wk, wmax = _lib.prescale_weights(w)
        wn = wk.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xn = x.permute(0, 2, 3, 1).contiguous()
    img = _lib.conv_image(wn, wn.shape[0], cin, k, stride, pad, _lib.PREC_SPLIT, wmax, deconv=bool(deconv))
    assert img is not None and img[1] == 1 and img[0].numel() == _lib.lib.hmvit_conv_gemm_image_bytes(wn.shape[0], wn.shape[1])

    def run(ring):
        y = torch.empty(N, Ho, Wo, cout, device="cuda")
        _lib.conv_range(xn, wmax, y, _stream())
        if ring:
            _lib.use_conv_image(img)
        _lib.check(_lib.lib.hmvit_conv2d(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W, cin, cout, 1 if deconv else k,
                                         1 if deconv else stride, 0 if deconv else pad, 1, cout, 0, deconv, 0, _lib.PREC_SPLIT, _stream()), "conv")
        return y
    y_ring, y_regs = run(True), run(False)
    assert torch.equal(y_ring, y_regs), float((y_ring - y_regs).abs().max())
    assert rel_max_err(y_ring.permute(0, 3, 1, 2).double(), ref) < 4e-6
    # a kind-0 image handed to a call that takes the generic kernel is ignored, not misread
    y = torch.empty_like(y_ring)
    _lib.conv_range(xn, wmax, y, _stream())
    _lib.use_conv_image(img[0], 0)
    _lib.check(_lib.lib.hmvit_conv2d(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W, cin, cout, 1 if deconv else k,
                                     1 if deconv else stride, 0 if deconv else pad, 1, cout, 0, deconv, 0, _lib.PREC_SPLIT, _stream()), "conv")
    assert torch.equal(y, y_regs)


@pytest.mark.parametrize("prec_name", ["split", "f16"])
@pytest.mark.parametrize("N,cin,cout,H,W", [(5, 384, 256, 64, 64),      # the shrink header's strided layer (scaled down)
                                            (4, 64, 64, 50, 38),       # narrow channel tile, odd sizes: ragged output tiles
                                            (3, 128, 136, 33, 65),     # ragged channel tile, odd input sizes
                                            (6, 256, 256, 32, 32)])    # first layer of a deep block
def test_conv3x3_stride2_ring_kernel(prec_name, N, cin, cout, H, W):
    """The stride-2 form of the ring kernel (17 x 33 input patch under 8 x 16 outputs, weights from the kind-0 image) against
    float64 and against the generic kernel on the same operands (other summation order: round-off-level agreement)."""
    from hmvit_amd import _lib
    prec = _lib.PREC_SPLIT if prec_name == "split" else _lib.PREC_F16
    dt = torch.float32 if prec_name == "split" else torch.float16
    torch.manual_seed(N * cin + cout + 11)
    x = torch.randn(N, cin, H, W, device="cuda").to(dt)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).to(dt)
    b = torch.randn(cout, device="cuda")
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), 2, 1))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xn = x.permute(0, 2, 3, 1).contiguous()
    wmax = 0.0
    wk = w.float()
    if prec_name == "split":
        wk, wmax = _lib.prescale_weights(wk)
    wn = wk.permute(0, 2, 3, 1).reshape(cout, -1).to(dt).contiguous()
    img = _lib.conv_image(wn, cout, cin, 3, 2, 1, prec, wmax)
    assert img is not None and img[1] == 0

    def run(ring):
        y = torch.empty(N, Ho, Wo, cout, device="cuda", dtype=dt)
        if prec_name == "split":
            _lib.conv_range(xn, wmax, y, _stream())
        if ring:
            _lib.use_conv_image(img)
        # upsample2 bit 2: the stride-2 ring kernel also for the shapes where the library prefers the generic one (Cin < 256)
        _lib.check(_lib.lib.hmvit_conv2d_ex(xn.data_ptr(), wn.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, H, W, cin, cout, 3, 2, 1, 1,
                                            4 if ring else 0, 0, prec, _stream()), "conv")
        return y
    y_ring, y_gen = run(True), run(False)
    tol = 4e-6 if prec_name == "split" else 1.5e-3
    assert rel_max_err(y_ring.permute(0, 3, 1, 2).double(), ref) < tol
    assert rel_max_err(y_ring.double(), y_gen.double()) < tol
