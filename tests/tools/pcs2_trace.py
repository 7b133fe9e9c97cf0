"""Cycle trace of workgroup 0 of k_attention_pcs2 (split precision) inside the real block forward at cfg2: compute wave 0 and
loader wave 0 stamp every 32-key step (csrc/attn.hip PC2_TRACE; probe build: `make PROBE=1`).  WHICH=grid|local picks the
traced launch (the block runs local then grid; both write the same buffer, so the local launch is traced by zeroing after it ...
here simply: the LAST launch that ran = the dilated-grid stage; WHICH=local runs a block with num stages cut by env)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from hmvit_amd import synthetic as S

cfg = S.make_config(256, 8, 5, voxel=0.4, downsample=1)
torch.manual_seed(0)
blk = hmvit_amd.HeteroFusionBlock(cfg["hetero_fusion_block"])
blk.precision = "split"
blk = blk.cuda().eval()
scene = [t.cuda() for t in S.synthetic_scene(5, 256, 200, 704, [1] * 5, seed=1)]
y = blk(*scene); torch.cuda.synchronize()
tr = torch.zeros(8192, dtype=torch.int64, device="cuda")
os.environ["HMVIT_ATTN_TRACE"] = hex(tr.data_ptr())
y = blk(*scene); torch.cuda.synchronize()
t = tr.cpu()[4096:4096 + 2048].reshape(2, 64, 16)
c, l = t[0], t[1]
print("compute wave 0: step | tiles 0-1, tiles 2-3, rest | barrier wait | step total")
for i in range(1, 40):
    if int(c[i, 0]) == 0: break
    a = [int(c[i, k]) - int(c[i, 0]) for k in range(5)]
    if int(c[i, 1]) == 0:      # skipped (no visible key)
        print(f"{i:3d}  (no math)            rest {int(c[i,3]) - int(c[i,0]):6d} | wait {int(c[i,4]) - int(c[i,3]):6d}"); continue
    print(f"{i:3d}  {a[1]:6d} {a[2] - a[1]:6d}  rest {a[3] - a[2]:6d} | wait {a[4] - a[3]:6d} | {int(c[i,4]) - int(c[i-1,4]):6d}")
print("compute wave 0, item boundaries: step | epilogue (stores) | [next item] fetch+bits, wait Q, split Q, request next Q")
for i in range(1, 40):
    if int(c[i, 5]) == 0: continue
    e = int(c[i - 1, 3]) - int(c[i - 1, 10]) if int(c[i - 1, 10]) else -1
    print(f"{i:3d}  epilogue {e:6d} | barrier->top {int(c[i,5]) - int(c[i-1,4]):6d}  fetch+bits {int(c[i,6]) - int(c[i,5]):6d}  waitQ {int(c[i,7]) - int(c[i,6]):6d}"
          f"  splitQ {int(c[i,8]) - int(c[i,7]):6d}  requestQ {int(c[i,9]) - int(c[i,8]):6d}  ->step {int(c[i,0]) - int(c[i,9]):6d}")
print("loader wave 0: step | batch reads+blend p0, request p0, blend p1, request p1, ... | tail | barrier wait | step total")
for i in range(1, 40):
    if int(l[i, 0]) == 0: break
    d = [int(l[i, k]) - int(l[i, k - 1]) for k in range(1, 11)]
    print(f"{i:3d}  " + " ".join(f"{v:5d}" for v in d[:8]) + f" | {d[8]:5d} | wait {d[9]:6d} | {int(l[i,10]) - int(l[i-1,10]):6d}")
