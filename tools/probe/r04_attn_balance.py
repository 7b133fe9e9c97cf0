"""Probe build only (-DHMVIT_PROBE): when does each of the 256 persistent workgroups of the split attention kernel run out of items?
The item partition is static (workgroup b walks its own strided share of the item list), so the launch lasts as long as its slowest
workgroup.  Prints, for the last attention launch of a HeteroFusionBlock forward at cfg2 (dilated grid, 5 egos) and - with `local` -
for the local-window launch: mean / min / max busy time of the workgroups and the time the mean workgroup idles at the end."""
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
    tr = torch.zeros(2 * 256, dtype=torch.int64, device="cuda")
    os.environ["HMVIT_ATTN_TRACE"] = hex(tr.data_ptr())
    y = blk(*scene); torch.cuda.synchronize()
t = tr.cpu().reshape(256, 2).double()
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0        # us
busy = end - start
print(f"last attention launch (dilated grid): launch {float(end.max()):.0f} us; workgroup busy time mean {float(busy.mean()):.0f} us, "
      f"min {float(busy.min()):.0f}, max {float(busy.max()):.0f}; finish times: min {float(end.min()):.0f}, mean {float(end.mean()):.0f}, "
      f"max {float(end.max()):.0f} -> the mean workgroup idles {float(end.max() - end.mean()):.0f} us ({100 * float(1 - end.mean() / end.max()):.1f} % of the launch)")
per_xcd = [float(end[i::8].mean()) for i in range(8)]
print("mean finish time per XCD (workgroup b on XCD b % 8):", [round(v) for v in per_xcd])
