"""Are the FAX branch's uniform 2-5e-3 gradient differences ReLU-mask flips at the end of the decoder?  Count the positions where the
HIP training forward and the float64 oracle disagree on (output > 0), and the smallest positive outputs on both sides."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd
from oracle import fax_oracle as FO, camera_oracle as CAM, cvt_oracle as CO
cfg = FO.make_camera_config(image=64)
cfg["fax"]["self_attn"]["dropout"] = 0.0
torch.manual_seed(9)
net = hmvit_amd.FaxCameraEncoder(cfg, precision="f32")
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.6, 1.4); m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
        if isinstance(m, torch.nn.LayerNorm):
            m.weight.normal_(1, 0.1); m.bias.normal_(0, 0.1)
net.set_return_features()
sd = {k: (v.double().clone() if v.is_floating_point() else v.clone()) for k, v in net.state_dict().items()}
net = net.cuda().train()
batch = CAM.synthetic_batch(2, CAM.make_config(image=64), seed=10)
with CO.batch_statistics(), torch.no_grad():
    ref = FO.fax_camera_encoder({k: v.double() for k, v in batch.items()}, sd, cfg)
for run in range(3):
    out = net({k: v.cuda() for k, v in batch.items()}).detach().cpu().double()
    mism = ((out > 0) != (ref > 0))
    print(f"run {run}: output {tuple(out.shape)}, mask mismatches {int(mism.sum())} of {out.numel()}; "
          f"|values| at the mismatches: {[f'{float(v):.1e}' for v in torch.maximum(out, ref)[mism][:6]]}; "
          f"outputs below 1e-5 x max: {int(((ref > 0) & (ref < 1e-5 * ref.max())).sum())}")
