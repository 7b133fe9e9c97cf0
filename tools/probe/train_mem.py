"""Where the cfg2 training step's memory goes: torch's allocated bytes after the forward, inside the backward (hook on the first gradient)
and the sizes the C API plans (saved activations, backward workspace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import synthetic as S, train as T
cfg = S.make_config(256, 8, 5, voxel=0.4, downsample=1)
scene = [t.cuda() for t in S.synthetic_scene(5, 256, 200, 704, [1] * 5, seed=1)]
net = S.seeded_fusion(cfg, precision="f32", seed=0).cuda().train()
G = 2 ** 30
print(f"inputs + weights: {torch.cuda.memory_allocated() / G:.2f} GiB")
orig_empty = torch.empty
def spy(*a, **k):
    t = orig_empty(*a, **k)
    if t.is_cuda and t.numel() * t.element_size() > 2 ** 27:
        print(f"  torch.empty {t.numel() * t.element_size() / G:6.2f} GiB  (allocated now {torch.cuda.memory_allocated() / G:.2f})")
    return t
torch.empty = spy
y = net(*scene)
print(f"after forward: {torch.cuda.memory_allocated() / G:.2f} GiB, peak {torch.cuda.max_memory_allocated() / G:.2f}")
loss = (y - torch.randn_like(y)).pow(2).mean()
loss.backward()
torch.cuda.synchronize()
print(f"after backward: {torch.cuda.memory_allocated() / G:.2f} GiB, peak {torch.cuda.max_memory_allocated() / G:.2f}")
