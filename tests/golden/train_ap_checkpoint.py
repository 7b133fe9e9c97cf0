"""Build-container tool: trains the small checkpoint fixture tests/golden/ap_checkpoint.npz so that the AP@0.7 replay is not
vacuous (an untrained detector has AP@0.7 = 0 on both sides).

Model = the g9 LiDAR-only HM-ViT config (tests/golden/model_fixture.py: 64x48 pillar canvas -> 12x16 BEV, window 4, 3 agents)
evaluated by the CPU oracle, which is plain differentiable torch.  Everything keeps its seeded random weights except
`mlp_head` (LiDAR type), the LiDAR cls / reg heads and the affine parameters of the LiDAR decoder's BatchNorms (~137 k floats,
replay_scenes.is_trained) -- the frozen PointPillar + H3GAT stack acts as a random feature extractor, which is enough for the
easy synthetic scenes (axis-aligned vehicles of the anchor size) and keeps the fixture small.  With so few trainable floats the
fixture MEMORISES its 64 training scenes (AP@0.7 97.5 there, 0.3 on held-out scenes); the AP replay therefore replays the fitted
scenes -- its purpose is a detector with true positives at IoU 0.7 on both sides of the comparison, not generalisation.  Labels come from the REFERENCE's
own VoxelPostprocessor.generate_label (voxel_postprocessor.py:74-194; its Cython bbox_overlaps restated in numpy with the same
+1 convention, box_overlaps.pyx:17-57) and the loss is the reference's PointPillarLoss (loss/point_pillar_loss.py:68-142).
Usage: python tests/golden/train_ap_checkpoint.py   (about 10 minutes on 8 cores)"""
import json
import os
import re
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


def bbox_overlaps(boxes, query):
    """box_overlaps.pyx:17-57 in numpy (areas and intersections with the legacy +1)."""
    iw = np.minimum(boxes[:, None, 2], query[None, :, 2]) - np.maximum(boxes[:, None, 0], query[None, :, 0]) + 1
    ih = np.minimum(boxes[:, None, 3], query[None, :, 3]) - np.maximum(boxes[:, None, 1], query[None, :, 1]) + 1
    ab = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    aq = (query[:, 2] - query[:, 0] + 1) * (query[:, 3] - query[:, 1] + 1)
    inter = np.where((iw > 0) & (ih > 0), iw * ih, 0.0)
    return (inter / (ab[:, None] + aq[None] - inter)).astype(np.float32)


from oracle import decoder_oracle as DO  # noqa: E402
from oracle import hmvit_oracle as O  # noqa: E402
from oracle import pointpillar_oracle as PO  # noqa: E402
from oracle import postprocess_oracle as PPO  # noqa: E402
from oracle import voxelizer_oracle as VO  # noqa: E402
from model_fixture import model_config, model_state_dict  # noqa: E402
import replay_scenes as RS  # noqa: E402

_stub("shapely"); _stub("shapely.geometry", Polygon=PPO.Polygon)
_stub("opencood.visualization"); _stub("opencood.visualization.vis_utils")
_stub("opencood.utils.box_overlaps", bbox_overlaps=bbox_overlaps)
_stub("cv2"); _stub("mmcv", Config=object, DictAction=object)
gt_line = [l for l in open("/root/reference/opencood/data_utils/datasets/__init__.py") if l.startswith("GT_RANGE")][0]
_stub("opencood.data_utils.datasets", GT_RANGE=json.loads(re.search(r"\[.*?\]", gt_line).group(0)))
from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor  # noqa: E402
from opencood.loss.point_pillar_loss import PointPillarLoss  # noqa: E402


def features(cfg, sd, clouds, pw):
    """Frozen part: pillariser -> PointPillar -> the H3GAT blocks -> ego row (1, 1, H, W, C), no mlp_head yet."""
    la = cfg["lidar"]
    L = len(clouds)
    vox = [VO.point_to_voxel(c, la["voxel_size"], la["lidar_range"], 32, 70000) for c in clouds]
    vf = torch.from_numpy(np.concatenate([v[0] for v in vox]))
    vc = torch.from_numpy(np.concatenate([np.concatenate([np.full((len(v[1]), 1), i, np.int32), v[1]], 1) for i, v in enumerate(vox)]))
    vn = torch.from_numpy(np.concatenate([v[2] for v in vox]))
    lsd = {k[len("lidar_encoder."):]: v for k, v in sd.items() if k.startswith("lidar_encoder.")}
    feats = PO.point_pillar_features(vf, vc, vn, lsd, la, L)
    fsd = {k[len("fusion_net."):]: v for k, v in sd.items() if k.startswith("fusion_net.")}
    x = feats[None]
    mode = torch.ones(1, L, dtype=torch.int64)
    fc = cfg["hetero_fusion"]
    for _ in range(fc["num_iters"]):
        x = O.hetero_fusion_block(x, pw, mode, torch.tensor([L]), torch.ones(1, L, dtype=torch.int64), fsd, "hetero_fusion_block",
                                  fc["hetero_fusion_block"])
    return x[:, :1].permute(0, 1, 3, 4, 2).contiguous()


def head(cfg, sd, ego):
    """Trainable tail: mlp_head -> HeteroDecoder -> psm, rm.  ego (B, 1, H, W, C)."""
    B = ego.shape[0]
    mode = torch.ones(B, 1, dtype=torch.int64)
    fsd = {k[len("fusion_net."):]: v for k, v in sd.items() if k.startswith("fusion_net.")}
    y = O.hetero_feed_forward(ego, mode, fsd, "mlp_head")[:, 0].permute(0, 3, 1, 2)
    dsd = {k: v for k, v in sd.items() if k.startswith("decoder.")}
    return DO.hetero_decoder(y.unsqueeze(1), mode, dsd, cfg["hetero_decoder"], prefix="decoder")


def main():
    torch.manual_seed(0)
    cfg = model_config()
    sd = model_state_dict(cfg, RS.CKPT_SEED_WEIGHTS)
    la = cfg["lidar"]
    nx, ny = la["point_pillar_scatter"]["grid_size"][:2]
    params = PPO.make_params(W=nx // 2, H=ny // 2)
    params["anchor_args"]["cav_lidar_range"] = la["lidar_range"]
    pp = VoxelPostprocessor(params, train=True)
    anchors = pp.generate_anchor_box()

    def dataset(seed, n):
        rs = np.random.RandomState(seed)
        egos, labels, gts = [], [], []
        for _ in range(n):
            clouds, pw, boxes, gt = RS.make_scene(rs, la["lidar_range"])
            with torch.no_grad():
                egos.append(features(cfg, sd, clouds, pw))
            gt_pad = np.zeros((100, 7), np.float32)
            gt_pad[:len(boxes)] = boxes
            mask = np.zeros(100)
            mask[:len(boxes)] = 1
            labels.append(pp.generate_label(gt_box_center=gt_pad, anchors=anchors, mask=mask))
            gts.append(gt)
        lab = {k: torch.from_numpy(np.stack([l[k] for l in labels])).float() for k in labels[0]}
        return torch.cat(egos), lab, gts

    t0 = time.time()
    ego_tr, lab_tr, _ = dataset(RS.TRAIN_SCENE_SEED, 64)
    ego_ev, _, gt_ev = dataset(RS.EVAL_SCENE_SEED, 24)
    print(f"features of 88 scenes: {time.time() - t0:.0f} s; positives per scene {float(lab_tr['pos_equal_one'].sum()) / 64:.1f}")

    train_keys = [k for k in sd if RS.is_trained(k)]
    for k in train_keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    crit = PointPillarLoss({"cls_weight": 1.0, "reg": 2.0})
    opt = torch.optim.Adam([sd[k] for k in train_keys], lr=3e-3)
    n_it = 2000
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, n_it, eta_min=1e-4)
    for it in range(n_it):
        opt.zero_grad()
        psm, rm = head(cfg, sd, ego_tr)
        loss = crit({"psm": psm, "rm": rm}, lab_tr)
        loss.backward()
        opt.step()
        sched.step()
        if it % 100 == 0 or it == n_it - 1:
            print(it, float(loss), float(crit.loss_dict["reg_loss"]), float(crit.loss_dict["conf_loss"]), flush=True)

    with torch.no_grad():
        psm, rm = head(cfg, sd, ego_ev)
    stat = {t: {"tp": [], "fp": [], "gt": 0} for t in (0.3, 0.5, 0.7)}
    for b in range(len(gt_ev)):
        boxes, scores = PPO.post_process(params, [{"psm": psm[b:b + 1].numpy(), "rm": rm[b:b + 1].numpy(), "anchor_box": anchors,
                                                   "transformation_matrix": None}])
        for t in stat:
            PPO.caluclate_tp_fp(boxes, scores, gt_ev[b], stat, t)
    print({t: round(100 * PPO.calculate_ap(stat, t)[0], 2) for t in stat})
    out = {k: sd[k].detach().numpy() for k in train_keys}
    np.savez_compressed(os.path.join(HERE, "ap_checkpoint.npz"), **out)
    print("ap_checkpoint.npz:", os.path.getsize(os.path.join(HERE, "ap_checkpoint.npz")) // 1024, "KiB,", sum(v.size for v in out.values()), "floats")


if __name__ == "__main__":
    main()
