"""Cycle trace of workgroup 0 of k_attention_patch16 (HMVIT_PATCH_ATTENTION=2, probe library): waves 0 and 13 stamp every step."""
import os, sys
os.environ["HMVIT_PATCH_ATTENTION"] = "2"
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
names = ["describe", "waitK", "blendK", "reqV", "logits01", "waitV", "blendV", "reqK", "rest"]
for w, wave in ((0, 0), (1, 13)):
    c = t[w]
    print(f"wave {wave}: step | " + " ".join(f"{n:>8s}" for n in names) + " | step total | [prologue] [merge+epilogue]")
    for i in range(0, 40):
        if int(c[i, 0]) == 0: break
        top = int(c[i, 11]) if int(c[i, 11]) else int(c[i, 0])
        d = [int(c[i, 1]) - top] + [int(c[i, k]) - int(c[i, k - 1]) for k in range(2, 10)]
        extra = ""
        if int(c[i, 11]): extra += f" | pro: Qissue {int(c[i,12]) - int(c[i,0])} tables {int(c[i,13]) - int(c[i,12])} Qsplit {int(c[i,11]) - int(c[i,13])}"
        if int(c[i, 10]): extra += f" | merge+epi {int(c[i,10]) - int(c[i,9])}"
        nxt = int(c[i + 1, 0]) if i + 1 < 64 and int(c[i + 1, 0]) else 0
        print(f"{i:3d}  " + " ".join(f"{v:8d}" for v in d) + f" | {(nxt - int(c[i, 0])) if nxt else -1:7d}" + extra)
