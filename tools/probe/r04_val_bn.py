"""Why does the validation loss rise (28 -> 252 in profiles/r03_train.txt, run 2) while the training loss falls when the LiDAR encoder
trains too?  The encoder's BatchNorm layers carry the reference's momentum 0.01 (pillar_vfe.py:25,
base_bev_backbone.py:46): after the 11 steps of that run the running statistics have moved 1 - 0.99^11 = 10 % of the way from their
initial (0, 1) to the batch statistics the weights were trained against, and eval() normalises with them.  This probe trains the same
configuration and then evaluates the validation frames three ways: eval() (running statistics), train() (batch statistics, dropout on),
and eval() after the running statistics were replaced by their converged values (momentum 1 for one pass over the training frames)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hmvit_amd  # noqa: F401
from hmvit_amd import trainer as T


class A:
    epochs, frames, agents, grid, small, precision, val_frames = 2, 6, 5, [512, 192], False, "f32", 2
    camera_ratio, ego_mode, camera_image, train_camera_backbone, train_lidar_backbone, model_dir, seed = 0.0, "mixed", 64, False, True, None, 0


hypes = T.default_hypes(A.epochs)
cfg, model, pre, post, ds, val = T.build(A)
model = model.cuda()
res = T.train(model, ds, pre, hypes, val_dataset=val)
crit = T.create_loss(hypes)
dev = next(model.parameters()).device


def val_loss(train_mode):
    out = []
    for i in range(len(val)):
        model.train(train_mode)
        batch = T.to_batch(val[i], pre, dev)
        with torch.set_grad_enabled(train_mode):
            out.append(float(crit(model(batch), batch["label_dict"])))
    return sum(out) / len(out)


eval_loss = val_loss(False)
batch_loss = val_loss(True)
bns = [m for m in model.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)) and m.training is not None]
moved = [float((m.running_var - 1).abs().mean()) for m in bns if m.track_running_stats]
# converge the running statistics: cumulative average over one pass of the training frames (momentum None)
old = [(m, m.momentum) for m in bns]
for m in bns:
    m.reset_running_stats()
    m.momentum = None
for i in range(len(ds)):
    model.train()
    batch = T.to_batch(ds[i], pre, dev)
    model(batch)
for m, mom in old:
    m.momentum = mom
conv_loss = val_loss(False)
print(json.dumps({"train": res["epoch_loss"], "val_during_training": res["val_loss"], "val_eval_running_stats": eval_loss,
                  "val_batch_statistics": batch_loss, "val_eval_converged_running_stats": conv_loss,
                  "mean_abs_running_var_minus_1": sum(moved) / max(1, len(moved)), "steps": res["steps"]}))
