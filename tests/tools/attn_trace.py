"""Cycle trace of workgroup 0 of the persistent attention kernel inside the real block forward at cfg2 (visibility table on):
compute wave 0 stamps per chunk step (csrc/attn.hip PC_TRACE): done, barrier in, barrier out.  PART=grid|local picks the
stage that is traced (the block runs local then grid; the local stage is traced by running the block with the grid launch's
trace switched off)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import synthetic as S

cfg = S.make_config(256, 8, 5, voxel=0.4, downsample=1)
torch.manual_seed(0)
blk = hmvit_amd.HeteroFusionBlock(cfg["hetero_fusion_block"])
blk.precision = "f16"
blk = blk.cuda().eval()
scene = [t.cuda() for t in S.synthetic_scene(5, 256, 200, 704, [1] * 5, seed=1)]
y = blk(*scene); torch.cuda.synchronize()
tr = torch.zeros(64 * 8, dtype=torch.int64, device="cuda")
os.environ["HMVIT_ATTN_TRACE"] = hex(tr.data_ptr())
y = blk(*scene); torch.cuda.synchronize()
t = tr.cpu().reshape(64, 8)
vals = [int(v) for v in t.flatten() if int(v)]
base = min(vals)
print("last attention launch of the block (dilated grid stage, 5 egos): step | compute done, barrier in, barrier out | compute time, wait")
prev_out = None
for i in range(40):
    d, bi, bo = (int(t[i, 5]) - base, int(t[i, 6]) - base, int(t[i, 7]) - base)
    if int(t[i, 5]) == 0:
        break
    comp = d - prev_out if prev_out is not None else -1
    print(f"{i:3d} {d:9d} {bi:9d} {bo:9d} | compute {comp:6d}  store/misc {bi - d:5d}  wait {bo - bi:6d}")
    prev_out = bo
