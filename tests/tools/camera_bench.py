"""Times the CVT camera branch at the shipped size: 5 agents x 4 cameras of 512 x 512, ResNet-34, pyramid levels 1 and 3,
32 x 32 BEV queries of dim 128, decoder to (256, 128, 128).  A tuning aid, not the headline bench."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import camera_oracle as CAM

cfg = CAM.make_config(image=512, num_layers=34, dim=128, bev=256)
sd = CAM.random_state_dict(cfg, seed=1)
batch = {k: v.cuda() for k, v in CAM.synthetic_batch(5, cfg, seed=2).items()}
for prec in (sys.argv[1:] or ["f16", "f32"]):
    net = hmvit_amd.CvtCameraEncoder(cfg, precision=prec)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    y = net(batch); torch.cuda.synchronize()
    n = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = net(batch)
    e1.record(); torch.cuda.synchronize()
    print(f"CvtCameraEncoder {prec}: {e0.elapsed_time(e1) / n:.2f} ms per 5-agent call, out {tuple(y.shape)}")
