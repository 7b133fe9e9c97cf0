"""Per-phase cycle sums of k_ln_qkv for one workgroup (library built with -DCHAIN_PHASES); cfg2 shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hmvit_amd
from oracle import hmvit_oracle as O

cfg = O.make_config(256, 8, 5, voxel=0.4, downsample=4, num_iters=1)
sd = O.random_state_dict(cfg, seed=1)
net = hmvit_amd.HeteroFusion(cfg, precision="f16")
net.load_state_dict(sd)
net = net.cuda()
scene = [t.cuda() for t in O.synthetic_scene(5, 256, 200, 704, [1] * 5, seed=1)]
tr = torch.zeros(64, dtype=torch.int64, device="cuda")
os.environ["HMVIT_QKV_TRACE"] = hex(tr.data_ptr())
tf = torch.zeros(64, dtype=torch.int32, device="cuda")
os.environ["HMVIT_FFN_TRACE"] = hex(tf.data_ptr())
net(*scene); torch.cuda.synchronize()
f = tf.cpu().tolist()
if sum(f):
    fn = ["issue loads + first DMA", "wait loads + barrier", "out-proj loop (8 tiles)", "LayerNorm", "8 x FFN1 MFMA", "8 x wait + barrier",
          "8 x GELU + DMA issue", "8 x FFN2 MFMA", "8 x wait + barrier", "store issue"]
    tot = sum(f[:10])
    print("k_out_ffn wave 0 of one workgroup, total cycles:", tot)
    for n, v in zip(fn, f[:10]):
        print(f"  {n:28s} {v:9d}  {100.0 * v / max(tot, 1):5.1f}%")
t = tr.cpu().tolist()
names = ["issue x loads + first DMA", "wait loads (vmcnt 0)", "syncthreads + 24 x DMA issue", "LayerNorm + operands", "24 x MFMA tile", "24 x dma/store wait",
         "24 x store issue", "24 x barrier"]
tot = sum(t[:8])
print("total cycles of the workgroup's wave 0:", tot)
for n, v in zip(names, t[:8]):
    print(f"  {n:28s} {v:9d}  {100.0 * v / max(tot, 1):5.1f}%")
