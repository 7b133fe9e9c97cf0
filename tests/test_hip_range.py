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


@pytest.mark.parametrize("precision", ["split", "f32"])
@pytest.mark.parametrize("case", list(CASES))
def test_fusion_dynamic_range(precision, case):
    import hmvit_amd
    kw = dict(CASES[case])
    cfg = O.make_config(256, 8, 3)
    sd = scaled_state_dict(cfg, 1, kw.pop("wscale", 1.0))
    scene = stress_scene(**kw)
    truth = O.hetero_fusion(*scene, sd, cfg, dtype=torch.float64)
    ref32 = O.hetero_fusion(*scene, sd, cfg)
    noise = error_report(ref32, truth)
    net = hmvit_amd.HeteroFusion(cfg, precision=precision)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    y = net(*[t.cuda() for t in scene]).cpu()
    assert torch.isfinite(y).all(), f"{case}: non-finite output"
    e = error_report(y, truth)
    print(f"\nrange[{precision}:{case}] rel-max {e['rel_max']:.2e} rms-rel {e['rms_rel']:.2e} p99.9 {e['p999']:.2e}"
          f"   (fp32 oracle vs float64: {noise['rel_max']:.2e} / {noise['rms_rel']:.2e} / {noise['p999']:.2e})")
    assert e["rel_max"] < max(1e-4, 4 * noise["rel_max"])
    assert e["rms_rel"] < max(1e-4, 4 * noise["rms_rel"])
