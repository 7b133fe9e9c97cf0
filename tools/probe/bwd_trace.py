"""Probe build only (make PROBE=1): cycle stamps of one workgroup of k_attention_bwd inside the cfg2 training step."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import _lib, synthetic as S, train as T
cfg = S.make_config(256, 8, 5, voxel=0.4, downsample=1)
scene = [t.cuda() for t in S.synthetic_scene(5, 256, 200, 704, [1] * 5, seed=1)]
net = S.seeded_fusion(cfg, precision="f32", seed=0).cuda().train()
target = torch.randn(1, 256, 200, 704, device="cuda")
for _ in range(2):
    net.zero_grad()
    (net(*scene) - target).pow(2).mean().backward()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
fn = _lib.lib.hmvit_debug_bwd_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 256) == 0
t0 = min(buf[w * 32] for w in range(4))
print("wave  start  prologue | per chunk: gather(issue+stage)  barrier  products+stores  barrier ... | dq")
for w in range(4):
    r = [buf[w * 32 + k] for k in range(32)]
    line = f"{w}: {r[0]-t0:7d} {r[1]-r[0]:7d} |"
    prev = r[1]
    for c in range(5):
        g, b, pdone = r[2 + 3 * c], r[3 + 3 * c], r[4 + 3 * c]
        if g == 0: break
        line += f" g{g-prev:7d} b{b-g:6d} p{pdone-b:7d}"
        prev = pdone
    line += f" | dq {r[30]-prev:7d}  total {r[30]-r[0]:8d}"
    print(line)
