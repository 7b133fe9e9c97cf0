import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import os; os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import camera_oracle as CAM
from hmvit_amd.camera import CvtCameraEncoder
ccfg = CAM.make_config(image=64, num_layers=18)
ccfg["cvm"]["bev_embedding"].update(bev_height=32, bev_width=32)
net = CvtCameraEncoder(ccfg, precision="f32")
net.load_state_dict(CAM.random_state_dict(ccfg, seed=31), strict=False)
net = net.cuda().train()
batch = {k: v.cuda() for k, v in CAM.synthetic_batch(2, ccfg, seed=32).items()}
out = net(batch)
go = torch.randn(out.shape, generator=torch.Generator().manual_seed(33)).cuda()
(out * go).sum().backward()
def loss():
    return float((net(batch).detach().double() * go.double()).sum())
gen = torch.Generator(device="cuda").manual_seed(34)
base = [loss() for _ in range(3)]
print("repeat loss", base)
for name, p in net.named_parameters():
    if name not in ("encoder.encoder.conv1.weight", "encoder.encoder.layer1.0.conv2.weight", "encoder.encoder.layer3.0.conv1.weight", "decoder.layer_0.conv.weight", "cvm.cross_views.0.cross_attend.to_q.1.weight"):
        continue
    d = torch.randn(p.shape, device="cuda", generator=gen) * p.detach().abs().mean().clamp_min(1e-3)
    an = float((p.grad.double() * d.double()).sum())
    res = []
    for eps in (2e-2, 5e-3, 1e-3, 2e-4):
        with torch.no_grad():
            p.add_(eps * d); up = loss(); p.sub_(2 * eps * d); down = loss(); p.add_(eps * d)
        res.append((eps, (up - down) / (2 * eps)))
    print(name, an, res)
