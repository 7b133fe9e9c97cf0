"""Debug aid: where do the parameter-gradient errors of the fusion's HIP backward come from at larger map sizes?
Runs the backward twice (run-to-run spread) and compares with the CPU oracle's autograd for a few scene variants."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hmvit_amd
from oracle import hmvit_oracle as O


def hip_grads(cfg, sd, scene, gy):
    net = hmvit_amd.HeteroFusion(cfg, precision="f32")
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    x = scene[0].cuda().requires_grad_(True)
    y = net(x, *[t.cuda() for t in scene[1:]])
    saved = y.grad_fn.launch[-1].clone() if hasattr(y.grad_fn, "launch") else None
    (y * gy.cuda()).sum().backward()
    hip_grads.saved = saved
    return {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None}, x.grad.cpu().double(), y.detach().cpu()


def oracle_grads(cfg, sd, scene, gy):
    x, pw, mode, rl, mask = scene
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    x = x.clone().requires_grad_(True)
    y = O.hetero_fusion(x, pw, mode, rl, mask, sd, cfg)
    (y * gy).sum().backward()
    return {k: v.grad.double() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}, x.grad.double(), y.detach()


def run(tag, L, H, W, modes, ds, tx, ty, yaw=0.2, oracle=True):
    cfg = O.make_config(256, 8, L, voxel=0.4, downsample=ds)
    sd = O.random_state_dict(cfg, seed=45)
    scene = O.synthetic_scene(L, 256, H, W, modes, seed=46, tx_step=tx, ty_step=ty, yaw_step=yaw)
    gy = torch.randn(1, 256, H, W, generator=torch.Generator().manual_seed(47))
    g1, gx1, y1 = hip_grads(cfg, sd, scene, gy)
    s1 = hip_grads.saved
    g2, gx2, y2 = hip_grads(cfg, sd, scene, gy)
    s2 = hip_grads.saved
    if s1 is not None:
        a, b = s1.view(torch.float32), s2.view(torch.float32)
        diff = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
        nd = int(diff.sum())
        first = int(diff.nonzero()[0]) if nd else -1
        print(f"[{tag}] saved activations: {nd} of {a.numel()} floats differ between two forwards; first at float offset {first}"
              f" (stage record = {a.numel() // 4} floats approx)")
    gmax = max(float(g.abs().max()) for g in g1.values())
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-4 * gmax)
    for key in ("window_attention.q_linears.1.bias", "window_attention.k_linears.1.bias", "window_attention.v_linears.1.bias",
                "window_attention.v_linears.1.weight", "window_attention.a_linears.1.0.weight", "window_ffd.fn.net.1.0.weight",
                "grid_attention.q_linears.1.bias", "grid_attention.v_linears.1.bias"):
        k = "hetero_fusion_block." + key
        if k in g1:
            print(f"[{tag}]    spread {key}: {rel(g1[k], g2[k]):.1e}")
    spread = sorted(((rel(g1[k], g2[k]), k) for k in g1), reverse=True)[:3]
    print(f"[{tag}] run-to-run spread:", [(f"{v:.1e}", k.split('block.')[-1]) for v, k in spread], "forward equal:", bool(torch.equal(y1, y2)))
    if oracle:
        t0 = time.time()
        go, gxo, yo = oracle_grads(cfg, sd, scene, gy)
        err = sorted(((rel(g1[k], go[k]), k) for k in g1 if k in go and float(go[k].abs().max()) > 0), reverse=True)[:5]
        print(f"[{tag}] vs oracle ({time.time() - t0:.0f} s): fwd {float((y1 - yo).abs().max() / yo.abs().max()):.1e} dx {rel(gx1, gxo):.1e}",
              [(f"{v:.1e}", k.split('block.')[-1]) for v, k in err])


def determinism(tag, L, H, W, modes, ds, tx, ty, n=6):
    cfg = O.make_config(256, 8, L, voxel=0.4, downsample=ds)
    sd = O.random_state_dict(cfg, seed=45)
    scene = O.synthetic_scene(L, 256, H, W, modes, seed=46, tx_step=tx, ty_step=ty)
    gy = torch.randn(1, 256, H, W, generator=torch.Generator().manual_seed(47))
    g0, _, _ = hip_grads(cfg, sd, scene, gy)
    gmax = max(float(g.abs().max()) for g in g0.values())
    worst = {}
    for _ in range(n - 1):
        g, _, _ = hip_grads(cfg, sd, scene, gy)
        for k in g0:
            v = float((g[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-4 * gmax)
            worst[k] = max(worst.get(k, 0.0), v)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:4]
    print(f"[{tag}] max spread over {n} runs:", [(f"{v:.1e}", k.split('block.')[-1]) for k, v in top])


def compare_dumps(H, W):
    import numpy as np
    if not (os.path.exists("/tmp/dbg_dq_0.bin") and os.path.exists("/tmp/dbg_dq_1.bin")):
        return
    a = np.fromfile("/tmp/dbg_dq_0.bin", np.float32).reshape(-1, H, W, 256)
    b = np.fromfile("/tmp/dbg_dq_1.bin", np.float32).reshape(-1, H, W, 256)
    d = np.abs(a - b)
    bad = d > 1e-4 * np.abs(a).max()
    print(f"[dq dump] {int((a != b).sum())} elements differ, {int(bad.sum())} by more than 1e-4 of max; max |diff| {d.max():.3e} at max |a| {np.abs(a).max():.3e}")
    idx = np.argwhere(bad)
    if len(idx):
        rows = sorted({(int(i[0]), int(i[1]) // 8, int(i[2]) // 8, int(i[3]) // 32) for i in idx})
        print(f"[dq dump] (agent, window row, window col, head) of grossly different elements: {rows[:12]} ... {len(rows)} in all")
        i0 = idx[0]
        print("[dq dump] first:", i0, a[tuple(i0)], b[tuple(i0)])


if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b", "c", "d"]
    if "det" in which: determinism("det 64x176 L2", 2, 64, 176, [1, 1], 2, 6.0, -4.0)
    if "a" in which: run("32x88 L5 mixed", 5, 32, 88, [1, 0, 1, 1, 0], 2, 6.0, -4.0)
    if "b" in which:
        run("64x176 L2 same type", 2, 64, 176, [1, 1], 2, 6.0, -4.0, oracle=False)
        compare_dumps(64, 176)
    if "c" in which: run("64x176 L5 no motion", 5, 64, 176, [1, 0, 1, 1, 0], 2, 0.0, 0.0, yaw=0.0)
    if "d" in which: run("64x176 L5 mixed", 5, 64, 176, [1, 0, 1, 1, 0], 2, 6.0, -4.0)
