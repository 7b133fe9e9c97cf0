"""AP replay on procedurally generated scenes: the same random-weight LiDAR-only HM-ViT model and the same
scenes through (i) the CPU restatement end to end and (ii) the HIP pipeline end to end (pillariser -> PointPillar -> regroup
-> HeteroFusion -> HeteroDecoder -> VoxelPostprocessor), then the reference's AP arithmetic on both against the scenes'
ground-truth boxes.  No OPV2V data and no trained checkpoint exist here, so the absolute AP is that of an untrained detector;
what the replay shows is the DELTA between the two pipelines.  Usage: python tests/tools/ap_replay.py [--scenes N] [--precision f32|f16]
Prints one JSON line."""
import argparse, json, math, os, sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import hmvit_amd
from model_fixture import model_config, model_state_dict
from oracle import decoder_oracle as DO
from oracle import hmvit_oracle as O
from oracle import pointpillar_oracle as PO
from oracle import postprocess_oracle as PPO
from oracle import voxelizer_oracle as VO


def make_scene(rs, cfg, n_agents=3, n_obj=6):
    """Vehicles (3.9 x 1.6 x 1.56 m) in the ego frame, agents at small offsets; every agent sees points on the vehicle
    surfaces (in its own frame) plus ground clutter."""
    rng = cfg["lidar"]["lidar_range"]
    poses = [O.rigid(0.15 * i, 2.5 * i, -1.5 * i) for i in range(n_agents)]          # agent i -> world (= ego) frame
    boxes = []
    for _ in range(n_obj):
        boxes.append([rs.uniform(rng[0] + 3, rng[3] - 3), rs.uniform(rng[1] + 3, rng[4] - 3), -1.0, 1.56, 1.6, 3.9,
                      rs.uniform(-math.pi, math.pi)])
    boxes = np.array(boxes, np.float32)
    gt = PPO.boxes_to_corners_3d(boxes, "hwl")
    clouds = []
    for i in range(n_agents):
        pts = []
        for b in boxes:
            u = rs.uniform(-0.5, 0.5, (150, 3)) * np.array([b[5], b[4], b[3]])
            face = rs.randint(0, 3, 150)
            u[np.arange(150), face] = np.sign(u[np.arange(150), face]) * 0.5 * np.array([b[5], b[4], b[3]])[face]
            c, s = math.cos(b[6]), math.sin(b[6])
            xy = np.stack([u[:, 0] * c - u[:, 1] * s + b[0], u[:, 0] * s + u[:, 1] * c + b[1], u[:, 2] + b[2]], 1)
            pts.append(xy)
        ground = np.stack([rs.uniform(rng[0], rng[3], 1500), rs.uniform(rng[1], rng[4], 1500), rs.uniform(-2.6, -2.3, 1500)], 1)
        world = np.concatenate(pts + [ground]).astype(np.float64)
        Tinv = np.linalg.inv(poses[i].numpy())
        local = (Tinv[:3, :3] @ world.T).T + Tinv[:3, 3]
        clouds.append(np.concatenate([local, rs.uniform(0, 1, (len(local), 1))], 1).astype(np.float32))
    pw = O.pairwise_from_poses(poses, n_agents)[None].float()
    return clouds, pw, gt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--precision", default="f32")
    args = ap.parse_args()
    cfg = model_config()
    sd = model_state_dict(cfg, 91)
    largs = cfg["lidar"]
    pre_params = {"cav_lidar_range": largs["lidar_range"], "args": {"voxel_size": largs["voxel_size"], "max_points_per_voxel": 32,
                                                                    "max_voxel_train": 32000, "max_voxel_test": 70000}}
    net = hmvit_amd.BevformerPointPillarHetero(cfg, precision=args.precision)
    net.load_state_dict(sd, strict=False)
    net = net.cuda().eval()
    pre = hmvit_amd.SpVoxelPreprocessor(pre_params, train=False)
    nx, ny = largs["point_pillar_scatter"]["grid_size"][:2]
    params = PPO.make_params(W=nx // 2, H=ny // 2)      # anchors on the (ny / 4, nx / 4) head grid, stride 2 of W, H
    params["anchor_args"]["cav_lidar_range"] = largs["lidar_range"]
    pp = hmvit_amd.VoxelPostprocessor(params, train=False)
    anchors = pp.generate_anchor_box()
    thr = (0.3, 0.5, 0.7)
    stat = {k: {t: {"tp": [], "fp": [], "gt": 0} for t in thr} for k in ("hip", "cpu")}
    n_det = {"hip": 0, "cpu": 0}
    rs = np.random.RandomState(7)
    for _ in range(args.scenes):
        clouds, pw, gt = make_scene(rs, cfg)
        L = len(clouds)
        mode = torch.ones(1, L, dtype=torch.float64)
        rl = torch.tensor([L])
        # ---- HIP pipeline ----
        lidar = pre.collate_batch([pre.preprocess(c) for c in clouds])
        out = net({"mode": mode.cuda(), "record_len": rl.cuda(), "pairwise_t_matrix": pw.cuda(), "processed_lidar": lidar})
        if out["psm"].shape[2:] != anchors.shape[:2]:
            raise SystemExit(f"anchor grid {anchors.shape[:2]} != head grid {tuple(out['psm'].shape[2:])}")
        # a random-weight head fires everywhere: keep the 200 most confident anchors of the HIP run as the operating point
        params["target_args"]["score_threshold"] = float(torch.sigmoid(out["psm"]).flatten().kthvalue(out["psm"].numel() - 200).values)
        data = {"ego": {"anchor_box": torch.from_numpy(anchors), "transformation_matrix": torch.eye(4)}}
        hb, hs = pp.post_process(data, {"ego": {"psm": out["psm"], "rm": out["rm"]}})
        # ---- CPU oracle pipeline ----
        vox = [VO.point_to_voxel(c, largs["voxel_size"], largs["lidar_range"], 32, 70000) for c in clouds]
        vf = torch.from_numpy(np.concatenate([v[0] for v in vox]))
        vc = torch.from_numpy(np.concatenate([np.concatenate([np.full((len(v[1]), 1), i, np.int32), v[1]], 1) for i, v in enumerate(vox)]))
        vn = torch.from_numpy(np.concatenate([v[2] for v in vox]))
        lsd = {k[len("lidar_encoder."):]: v for k, v in sd.items() if k.startswith("lidar_encoder.")}
        feats = PO.point_pillar_features(vf, vc, vn, lsd, largs, L)
        fsd = {k[len("fusion_net."):]: v for k, v in sd.items() if k.startswith("fusion_net.")}
        fused = O.hetero_fusion(feats[None], pw, mode.int(), rl, torch.ones(1, L, dtype=torch.int64), fsd, cfg["hetero_fusion"])
        dsd = {k: v for k, v in sd.items() if k.startswith("decoder.")}
        psm, rm = DO.hetero_decoder(fused.unsqueeze(1), mode.int(), dsd, cfg["hetero_decoder"], prefix="decoder")
        cb, cs = PPO.post_process(params, [{"psm": psm.numpy(), "rm": rm.numpy(), "anchor_box": anchors, "transformation_matrix": None}])
        for t in thr:
            hmvit_amd.caluclate_tp_fp(hb, hs, torch.from_numpy(gt).cuda(), stat["hip"], t)
            PPO.caluclate_tp_fp(cb, cs, gt, stat["cpu"], t)
        n_det["hip"] += 0 if hb is None else len(hb)
        n_det["cpu"] += 0 if cb is None else len(cb)
    res = {"scenes": args.scenes, "precision": args.precision, "detections": n_det}
    for t in thr:
        a = {k: (PPO.calculate_ap(stat[k], t)[0] if stat[k][t]["tp"] else 0.0) for k in ("hip", "cpu")}
        res[f"AP@{t}"] = {"hip": round(100 * a["hip"], 3), "cpu_oracle": round(100 * a["cpu"], 3),
                          "delta_points": round(100 * abs(a["hip"] - a["cpu"]), 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
