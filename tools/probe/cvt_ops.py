"""Which torch-level operations the CVT camera encoder's forward still runs (the hmvit kernels are C calls and do not show): aten op
counts of one eval forward in the split mode, from torch.profiler - the copyBuffer / elementwise launches of r05_cvt_layers.sh."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import camera_oracle as CAM
cfg = CAM.make_config(image=512, num_layers=34, dim=128, bev=256)
sd = CAM.random_state_dict(cfg, seed=1)
batch = {k: v.cuda() for k, v in CAM.synthetic_batch(5, cfg, seed=2).items()}
net = hmvit_amd.CvtCameraEncoder(cfg, precision=sys.argv[1] if len(sys.argv) > 1 else "split")
net.load_state_dict(sd, strict=False)
net = net.cuda().eval()
for _ in range(2):
    net(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    net(batch)
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.cpu_parent is None or (e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::") and e.name.startswith("aten::")):
        c[e.name] += 1
for k, v in c.most_common(25):
    print(f"{v:5d}  {k}")
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=40, max_src_column_width=90)[:6000])
