"""Dynamic-range stress of the fp32-parity modes: the `split` mode forms every fp32 product from f16
hi / lo operand halves, and f16 has 5 exponent bits - |x| > 65504 overflows, |x| < 6e-5 goes subnormal.  These tests feed
inputs and weights far away from unit scale and hold the HIP path to the SAME network evaluated in float64 on the CPU
(`O.hetero_fusion(..., dtype=torch.float64)`: identical fp32 sampling geometry, double-precision arithmetic).  The bound is
`max(1e-4, 4 x the fp32 oracle's own distance from that truth)`: where the reference's fp32 arithmetic is itself noisy
(weights x 100: 3.7e-4) nobody can be held to 1e-4, everywhere else 1e-4 rel-max is the bar, and the rms-relative and
99.9-percentile element-wise figures are printed next to it."""
import pytest
import torch

from oracle import hmvit_oracle as O

pytestmark = pytest.mark.gpu


def error_report(y, truth):
    """rel-max (max |d| / max |ref|), rms-relative (rms d / rms ref) and the 99.9-percentile of the element-wise relative
    error |d| / max(|ref|, 1e-3 rms ref)."""
    y, truth = y.double(), truth.double()
    d = (y - truth).abs()
    rms = truth.pow(2).mean().sqrt()
    elem = (d / truth.abs().clamp_min(1e-3 * rms)).flatten()
    k = max(1, int(round(0.999 * elem.numel())))
    return dict(rel_max=float(d.max() / truth.abs().max()), rms_rel=float(d.pow(2).mean().sqrt() / rms),
                p999=float(elem.kthvalue(k).values))


def scaled_state_dict(cfg, seed, wscale):
    sd = O.random_state_dict(cfg, seed=seed)
    if wscale != 1.0:
        for k in sd:
            if ("linears" in k or ".fn.net." in k or k.startswith("mlp_head")) and sd[k].is_floating_point():
                sd[k] = sd[k] * wscale      # every Linear of the path (weights and biases); LayerNorm affine left alone
    return sd


def stress_scene(xscale=1.0, outliers=0, L=3, H=16, W=24, modes=(1, 0, 1)):
    x, pw, mode, rl, mask = O.synthetic_scene(L, 256, H, W, list(modes), seed=2, tx_step=4.0, ty_step=-3.0)
    x = x * xscale
    if outliers:
        g = torch.Generator().manual_seed(5)
        idx = torch.randint(0, x.numel(), (outliers,), generator=g)
        x.view(-1)[idx] = 1e5 * torch.sign(torch.randn(outliers, generator=g))
    return x, pw, mode, rl, mask


CASES = {
    "base": dict(),
    "x1e-3": dict(xscale=1e-3),
    "x1e3": dict(xscale=1e3),
    "x3e4": dict(xscale=3e4),
    "outliers_1e5": dict(outliers=40),
    "x3e4_outliers": dict(xscale=3e4, outliers=40),
    "w1e-3": dict(wscale=1e-3),
    "w1e-2": dict(wscale=1e-2),
    "w10": dict(wscale=10.0),
    "w1e2": dict(wscale=1e2),
    "w1e-3_x1e3": dict(wscale=1e-3, xscale=1e3),
}


# a size where every scheduling path is live (persistent attention, world-ordered schedule, reachability tables, 440 chain
# workgroups): 5 agents 10110, 64 x 176 (the float64 oracle needs ~1 min per case on the CPU)
BIG = dict(L=5, H=64, W=176, modes=(1, 0, 1, 1, 0))
BIG_CASES = {      # (two cases: the float64 oracle at this size is a minute of CPU each)
    "big_x3e4_outliers": dict(BIG, xscale=3e4, outliers=400),
    "big_w1e-3_x1e3": dict(BIG, wscale=1e-3, xscale=1e3),
}
CASES.update(BIG_CASES)
_truth_cache = {}


def _truth(case):
    """float64 truth + the fp32 oracle's own distance from it; computed once per case for all precisions (`case` is the
    slowest-varying parameter below, and the cache keeps the current case only).  The float64 run takes its token arithmetic
    to the GPU through torch (`hetero_fusion(device="cuda")`: sampling geometry and visibility masks stay on the CPU) - at
    5 agents / 64 x 176 the CPU needs a minute per case, which the suite's budget on the driver does not have (VERDICT r4
    item 1); `test_float64_yardstick_is_the_same_on_both_devices` holds the two placements together.  The fp32 run that
    measures the reference arithmetic's own noise stays on the CPU for the small cases."""
    if case not in _truth_cache:
        kw = dict(CASES[case])
        cfg = O.make_config(256, 8, kw.get("L", 3))
        sd = scaled_state_dict(cfg, 1, kw.pop("wscale", 1.0))
        scene = stress_scene(**kw)
        truth = O.hetero_fusion(*scene, sd, cfg, dtype=torch.float64, device="cuda").cpu()
        noise = error_report(O.hetero_fusion(*scene, sd, cfg, device="cuda" if case in BIG_CASES else None).cpu(), truth)
        _truth_cache.clear()        # one entry at a time: the big cases hold 100 MB each
        _truth_cache[case] = (cfg, sd, scene, truth, noise)
    return _truth_cache[case]


@pytest.mark.parametrize("precision", ["split", "mixed", "f32"])     # innermost decorator varies slowest: keep `case` below it
@pytest.mark.parametrize("case", list(CASES))
def test_fusion_dynamic_range(precision, case):
    """`mixed` (split products in the Linears, attention operands rounded once to f16) is held to the same bound wherever the
    scale of the operands is all that changes: its static power-of-two scales must keep the f16 planes in range.  Where the
    WEIGHTS grow (w10, w1e2) the logits grow with their square and the softmax sharpens: rounding Q / K' to f16 then costs
    2^-11 of a logit of several hundred - the data dependence that keeps `mixed` from being the headline.  Those
    cases must stay finite and within 0.1; the measured figures are printed (6.9e-4 / 3.9e-2 / 5.0e-2 in round 4)."""
    import hmvit_amd
    cfg, sd, scene, truth, noise = _truth(case)
    net = hmvit_amd.HeteroFusion(cfg, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    y = net(*[t.cuda() for t in scene]).cpu()
    assert torch.isfinite(y).all(), f"{case}: non-finite output"
    e = error_report(y, truth)
    print(f"\nrange[{precision}:{case}] rel-max {e['rel_max']:.2e} rms-rel {e['rms_rel']:.2e} p99.9 {e['p999']:.2e}"
          f"   (fp32 oracle vs float64: {noise['rel_max']:.2e} / {noise['rms_rel']:.2e} / {noise['p999']:.2e})")
    if precision == "mixed" and CASES[case].get("wscale", 1.0) >= 10.0:
        assert e["rel_max"] < 0.1
        return
    assert e["rel_max"] < max(1e-4, 4 * noise["rel_max"])
    assert e["rms_rel"] < max(1e-4, 4 * noise["rms_rel"])


def test_float64_yardstick_is_the_same_on_both_devices():
    """The float64 oracle evaluated with its token arithmetic on the GPU (torch / rocBLAS in double precision) and on the CPU:
    same sampling positions, same visibility masks (both always from the CPU), so the two differ by double-precision round-off
    only - 1e-12 rel-max is asserted (measured 1e-15).  This is what allows the big cases to take their truth from the GPU."""
    cfg = O.make_config(256, 8, 3)
    sd = scaled_state_dict(cfg, 1, 1.0)
    scene = stress_scene()
    on_cpu = O.hetero_fusion(*scene, sd, cfg, dtype=torch.float64)
    on_gpu = O.hetero_fusion(*scene, sd, cfg, dtype=torch.float64, device="cuda").cpu()
    err = float((on_cpu - on_gpu).abs().max() / on_cpu.abs().max())
    print(f"\nfloat64 oracle, CPU vs GPU placement of the token arithmetic: rel-max {err:.2e}")
    assert err < 1e-12


# ---- convolutional modules in the split mode: PointPillar encoder and HeteroDecoder ----
def _report(tag, y, truth, ref32):
    noise, e = error_report(ref32, truth), error_report(y, truth)
    print(f"\nrange[{tag}] rel-max {e['rel_max']:.2e} rms-rel {e['rms_rel']:.2e} p99.9 {e['p999']:.2e}"
          f"   (fp32 oracle vs float64: {noise['rel_max']:.2e} / {noise['rms_rel']:.2e} / {noise['p999']:.2e})")
    assert torch.isfinite(y).all(), f"{tag}: non-finite output"
    assert e["rel_max"] < max(1e-4, 4 * noise["rel_max"])
    assert e["rms_rel"] < max(1e-4, 4 * noise["rms_rel"])


@pytest.mark.parametrize("precision", ["split", "f32"])
@pytest.mark.parametrize("case,xscale,wscale", [("base", 1.0, 1.0), ("x1e-3", 1e-3, 1.0), ("x1e3", 1e3, 1.0), ("x3e4", 3e4, 1.0),
                                                ("w1e-3", 1.0, 1e-3), ("w1e2", 1.0, 1e2), ("w1e-3_x3e4", 3e4, 1e-3)])
def test_decoder_dynamic_range(precision, case, xscale, wscale):
    """HeteroDecoder (four 3 x 3 convolutions + BatchNorm + ReLU, two 1 x 1 heads) on inputs / convolution weights far from unit
    scale (the BatchNorm statistics stay, so the scale change is not normalised away)."""
    import numpy as np
    import hmvit_amd
    from oracle import decoder_oracle as DO
    params = DO.make_params()
    sd = DO.random_state_dict(params, 5)
    for k in sd:
        if k.endswith(".weight") and sd[k].dim() == 4:
            sd[k] = sd[k] * wscale
    x = torch.from_numpy(np.random.RandomState(11).standard_normal((3, 1, 256, 12, 10)).astype(np.float32)) * xscale
    mode = torch.tensor([[1], [0], [1]])
    truth = DO.hetero_decoder(x, mode, sd, params, dtype=torch.float64)
    ref32 = DO.hetero_decoder(x, mode, sd, params)
    net = hmvit_amd.HeteroDecoder(params, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    psm, rm = net(x.cuda(), mode.cuda(), use_upsample=False)
    _report(f"decoder:{precision}:{case}:psm", psm.cpu(), truth[0], ref32[0])
    _report(f"decoder:{precision}:{case}:rm", rm.cpu(), truth[1], ref32[1])


@pytest.mark.parametrize("precision", ["split", "f32"])
@pytest.mark.parametrize("case,wscale,pscale", [("base", 1.0, 1.0), ("w1e-2", 1e-2, 1.0), ("w30", 30.0, 1.0), ("intensity_1e4", 1.0, 1e4)])
def test_pointpillar_dynamic_range(precision, case, wscale, pscale):
    """PointPillar encoder with its convolution weights (not the BatchNorm affine / statistics) scaled, and with the point
    intensity channel scaled (raw LiDAR intensities are not always in [0, 1])."""
    import hmvit_amd
    from oracle import pointpillar_oracle as PO
    nx, ny = 64, 48
    args = PO.make_args(nx, ny)
    sd = PO.random_state_dict(args, 9)
    for k in sd:
        if k.endswith(".weight") and sd[k].dim() == 4:
            sd[k] = sd[k] * wscale
    vf, vc, vn = PO.synthetic_pillars(2, 400, nx, ny, args, 4)
    vf = vf.clone()
    vf[:, :, 3] *= pscale
    truth = PO.point_pillar_features(vf, vc, vn, sd, args, 2, dtype=torch.float64)
    ref32 = PO.point_pillar_features(vf, vc, vn, sd, args, 2)
    net = hmvit_amd.PointPillar(args, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    net.set_return_features()
    batch = {"processed_lidar": {"voxel_features": vf.cuda(), "voxel_coords": vc.cuda(), "voxel_num_points": vn.cuda()}}
    _report(f"pointpillar:{precision}:{case}", net(batch).cpu(), truth, ref32)
