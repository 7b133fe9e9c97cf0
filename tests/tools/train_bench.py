"""Times one training step of the fusion (HIP forward with dropout + HIP backward + AdamW) at bench.py's configurations:
configs[4] of BASELINE.json is the train loop; this is its fusion part on one GPU.  Not the headline bench; a measuring aid.
    python tests/tools/train_bench.py [cfg2|native] [steps] [recompute bits]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import synthetic as S, train as T

name = sys.argv[1] if len(sys.argv) > 1 else "native"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
recompute = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # HmvitFusionTrainDesc::recompute bits (1: FFN pre-activations, 3: + queries)
c = {"cfg2": dict(L=5, C=256, H=200, W=704, window=8, modes=[1] * 5, voxel=0.4, downsample=1),
     "native": dict(L=5, C=256, H=128, W=128, window=8, modes=[1, 0, 1, 1, 0], voxel=0.4, downsample=4)}[name]
cfg = S.make_config(c["C"], c["window"], c["L"], voxel=c["voxel"], downsample=c["downsample"])
scene = [t.cuda() for t in S.synthetic_scene(c["L"], c["C"], c["H"], c["W"], c["modes"], seed=1)]
net = S.seeded_fusion(cfg, precision="f32", seed=0).cuda().train()
net.train_recompute = recompute
opt = T.make_optimizer(net.parameters())
target = torch.randn(1, c["C"], c["H"], c["W"], device="cuda")
def step():
    opt.zero_grad()
    y = net(*scene)
    loss = (y - target).pow(2).mean()
    loss.backward()
    opt.step()
    return loss
step(); torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t_f = t_b = 0.0
for _ in range(steps):
    opt.zero_grad()
    ev[0].record(); y = net(*scene); ev[1].record()
    loss = (y - target).pow(2).mean()
    ev[2].record(); loss.backward(); ev[3].record()
    opt.step()
    torch.cuda.synchronize()
    t_f += ev[0].elapsed_time(ev[1]); t_b += ev[2].elapsed_time(ev[3])
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"{name}: train step {ms:.1f} ms (forward {t_f / steps:.1f} ms, backward {t_b / steps:.1f} ms), "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, loss {float(loss.detach()) if (loss := step()) is not None else 0:.4f}")
