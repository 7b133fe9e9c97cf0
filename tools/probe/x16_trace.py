"""Probe build only (make PROBE=1): cycle stamps of one workgroup of the split tail kernel inside the real cfg2 forward."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import _lib, synthetic as S
prec = sys.argv[1] if len(sys.argv) > 1 else "split"
cfg = S.make_config(256, 8, 5, voxel=0.4, downsample=1)
scene = [t.cuda() for t in S.synthetic_scene(5, 256, 200, 704, [1, 1, 1, 1, 1], seed=1)]
net = S.seeded_fusion(cfg, precision=prec).cuda().eval()
with torch.no_grad():
    for _ in range(3):
        net(*scene)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
fn = _lib.lib.hmvit_debug_x16_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 1024) == 0
t0 = buf[0]
for grp in (0, 1):
    rows = [[buf[(grp * 64 + i) * 8 + k] for k in range(6)] for i in range(64)]
    print(f"group {'AB'[grp]}: step  begin  products  confirm  mid-barrier  rest+request  end-barrier  total")
    for i, r in enumerate(rows[:50]):
        if r[0] == 0: continue
        c2 = r[2] if r[2] else r[1]
        print(f"{i:3d} {r[0]-t0:8d} {r[1]-r[0]:8d} {c2-r[1]:8d} {r[3]-c2:8d} {r[4]-r[3]:10d} {r[5]-r[4]:10d} {r[5]-r[0]:8d}")
