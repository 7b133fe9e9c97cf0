"""Which torch (aten) operators the CVT camera branch still launches per forward, and from where: torch.profiler over one
CvtCameraEncoder forward (split), grouped by operator and by the first hm-vit_amd source line on the Python stack."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import camera_oracle as CAM
from torch.profiler import profile, ProfilerActivity

cfg = CAM.make_config(image=512, num_layers=34, dim=128, bev=256)
sd = CAM.random_state_dict(cfg, seed=1)
batch = {k: v.cuda() for k, v in CAM.synthetic_batch(5, cfg, seed=2).items()}
net = hmvit_amd.CvtCameraEncoder(cfg, precision=(sys.argv[1] if len(sys.argv) > 1 else "split"))
net.load_state_dict(sd, strict=False)
net = net.cuda().eval()
with torch.no_grad():
    for _ in range(2):
        net(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        net(batch)
        torch.cuda.synchronize()
by_site = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    site = next((s for s in (ev.stack or []) if "hm-vit_amd" in s), "?")
    by_site[(ev.name, site.split("hm-vit_amd/")[-1][:70])] += 1
    dur[(ev.name, site.split("hm-vit_amd/")[-1][:70])] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
print("top-level aten ops of one forward: count, device us, op, call site")
for k, n in by_site.most_common(60):
    print(f"{n:4d} {dur[k]:9.1f}  {k[0]:34s} {k[1]}")
print("total top-level aten ops:", sum(by_site.values()))
