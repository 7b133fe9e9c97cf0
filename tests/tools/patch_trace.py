"""Cycle trace of workgroup 0 of k_attention_patch (split precision, local stage) inside the real block forward at cfg2: waves 0 and 5
stamp every 32-key step (csrc/attn_patch.hpp PATCH_TRACE; probe build: HMVIT_LIB=tools/probe/lib_probe.so)."""
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
with torch.no_grad():
    y = blk(*scene); torch.cuda.synchronize()
    tr = torch.zeros(8192, dtype=torch.int64, device="cuda")
    os.environ["HMVIT_ATTN_TRACE"] = hex(tr.data_ptr())
    y = blk(*scene); torch.cuda.synchronize()
t = tr.cpu()[1024:1024 + 2048].reshape(2, 64, 16)
names = ["describe", "wait", "blendK", "blendV", "request", "math"]
slots = [1, 2, 3, 4, 5, 9]
for w, wave in ((0, 0), (1, 5)):
    c = t[w]
    print(f"wave {wave}: step | " + " ".join(f"{n:>8s}" for n in names) + " | step total | [prologue: B1, tables, Q] [epilogue]")
    for i in range(0, 48):
        if int(c[i, 0]) == 0: break
        top = int(c[i, 11]) if int(c[i, 11]) else int(c[i, 0])
        d = [int(c[i, 1]) - top] + [int(c[i, slots[k]]) - int(c[i, slots[k - 1]]) for k in range(1, len(slots))]
        extra = ""
        if int(c[i, 11]):
            extra += f" | pro: B1 {int(c[i,12]) - int(c[i,0])} tables {int(c[i,13]) - int(c[i,12])} Q {int(c[i,11]) - int(c[i,13])}"
        if int(c[i, 10]):
            extra += f" | epi {int(c[i,10]) - int(c[i,9])}"
        nxt = int(c[i + 1, 0]) if i + 1 < 64 and int(c[i + 1, 0]) else 0
        tot = (nxt - int(c[i, 0])) if nxt else -1
        print(f"{i:3d}  " + " ".join(f"{v:8d}" for v in d) + f" | {tot:7d}" + extra)
