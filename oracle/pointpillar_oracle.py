"""CPU oracle for the LiDAR BEV encoder (TEST INFRASTRUCTURE, not product code).

Plain-PyTorch fp32 restatement, in eval mode, of ``PointPillar.forward`` with
``return_features`` set (opencood/models/point_pillar.py:35-54): PillarVFE -> PointPillarScatter
-> BaseBEVBackbone -> DownsampleConv.  Takes the reference's ``state_dict`` (same key names).
Imported only by tests (and by nothing in the shipped package).

Parity status: PINNED by tests/golden/g7_pointpillar.npz (reference imported in the build
container by tests/golden/make_goldens.py).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def pillar_vfe(voxel_features: Tensor, voxel_num_points: Tensor, coords: Tensor, sd: Dict[str, Tensor],
               voxel_size, lidar_range, prefix: str = "pillar_vfe") -> Tensor:
    """PillarVFE.forward with one PFN layer, use_norm, use_absolute_xyz, no distance
    (sub_modules/pillar_vfe.py:105-146, 31-53).  voxel_features (Nv, 32, 4), coords (Nv, 4)
    [agent, z, y, x] -> (Nv, 64)."""
    vx, vy, vz = voxel_size
    x_off, y_off, z_off = vx / 2 + lidar_range[0], vy / 2 + lidar_range[1], vz / 2 + lidar_range[2]
    xyz = voxel_features[:, :, :3]
    mean = xyz.sum(1, keepdim=True) / voxel_num_points.to(xyz.dtype).view(-1, 1, 1)
    f_cluster = xyz - mean
    centre = torch.stack([coords[:, 3].to(xyz.dtype) * vx + x_off, coords[:, 2].to(xyz.dtype) * vy + y_off,
                          coords[:, 1].to(xyz.dtype) * vz + z_off], dim=-1)
    f_center = xyz - centre[:, None, :]
    feats = torch.cat([voxel_features, f_cluster, f_center], dim=-1)             # (Nv, 32, 10)
    n_pts = feats.shape[1]
    valid = voxel_num_points.view(-1, 1).int() > torch.arange(n_pts, dtype=torch.int).view(1, -1)
    feats = feats * valid.unsqueeze(-1).to(feats.dtype)                           # padded points zeroed AFTER augmentation
    p = f"{prefix}.pfn_layers.0"
    x = F.linear(feats, sd[f"{p}.linear.weight"])
    x = F.batch_norm(x.permute(0, 2, 1), sd[f"{p}.norm.running_mean"], sd[f"{p}.norm.running_var"],
                     sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], False, 0.0, 1e-3).permute(0, 2, 1)
    return F.relu(x).max(dim=1)[0]


def scatter(pillar_features: Tensor, coords: Tensor, n_agents: int, ny: int, nx: int) -> Tensor:
    """PointPillarScatter.forward (sub_modules/point_pillar_scatter.py:14-47): index z + y*nx + x."""
    C = pillar_features.shape[1]
    out = torch.zeros(n_agents, C, ny * nx, dtype=pillar_features.dtype)
    for b in range(n_agents):
        m = coords[:, 0] == b
        idx = (coords[m, 1] + coords[m, 2] * nx + coords[m, 3]).long()
        out[b][:, idx] = pillar_features[m].t()
    return out.view(n_agents, C, ny, nx)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"],
                        False, 0.0, 1e-3)


def bev_backbone(x: Tensor, sd: Dict[str, Tensor], cfg: dict, prefix: str = "backbone") -> Tensor:
    """BaseBEVBackbone.forward (backbones/base_bev_backbone.py:89-122) for upsample strides >= 1."""
    ups = []
    for i, (n_layers, stride) in enumerate(zip(cfg["layer_nums"], cfg["layer_strides"])):
        b = f"{prefix}.blocks.{i}"
        x = F.relu(_bn(F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[f"{b}.1.weight"], None, stride), sd, f"{b}.2"))
        for k in range(n_layers):
            x = F.relu(_bn(F.conv2d(x, sd[f"{b}.{4 + 3 * k}.weight"], None, 1, 1), sd, f"{b}.{5 + 3 * k}"))
        d = f"{prefix}.deblocks.{i}"
        us = cfg["upsample_strides"][i]
        ups.append(F.relu(_bn(F.conv_transpose2d(x, sd[f"{d}.0.weight"], None, us), sd, f"{d}.1")))
    return torch.cat(ups, dim=1)


def shrink_conv(x: Tensor, sd: Dict[str, Tensor], cfg: dict, prefix: str = "shrink_conv") -> Tensor:
    """DownsampleConv.forward (sub_modules/downsample_conv.py:32-51)."""
    for i, (stride, pad) in enumerate(zip(cfg["stride"], cfg["padding"])):
        p = f"{prefix}.layers.{i}.double_conv"
        x = F.relu(F.conv2d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], stride, pad))
        x = F.relu(F.conv2d(x, sd[f"{p}.2.weight"], sd[f"{p}.2.bias"], 1, 1))
    return x


def point_pillar_features(voxel_features, voxel_coords, voxel_num_points, sd, args, n_agents: int,
                          dtype: torch.dtype = torch.float32) -> Tensor:
    """PointPillar.forward up to ``spatial_features_2d`` (point_pillar.py:35-54).  dtype=torch.float64: the same layers in
    double precision (yardstick of the precision stress tests)."""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    nx, ny, nz = args["point_pillar_scatter"]["grid_size"]
    pf = pillar_vfe(voxel_features.to(dtype), voxel_num_points, voxel_coords, sd, args["voxel_size"], args["lidar_range"])
    canvas = scatter(pf, voxel_coords, n_agents, int(ny), int(nx))
    x = bev_backbone(canvas, sd, args["base_bev_backbone"])
    if "shrink_header" in args:
        x = shrink_conv(x, sd, args["shrink_header"])
    return x


# ---- seeded synthetic inputs / weights (numpy legacy stream, shared by goldens and tests) ----
def make_args(nx: int = 64, ny: int = 64, small: bool = True) -> dict:
    vs = [0.4, 0.4, 4]
    rng = [-nx * 0.2, -ny * 0.2, -3, nx * 0.2, ny * 0.2, 1]
    return {"voxel_size": vs, "lidar_range": rng, "anchor_number": 2, "cls_head_dim": 256,
            "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
            "point_pillar_scatter": {"num_features": 64, "grid_size": [nx, ny, 1]},
            "base_bev_backbone": {"layer_nums": [1, 2, 2] if small else [3, 5, 8], "layer_strides": [2, 2, 2],
                                  "num_filters": [64, 128, 256], "upsample_strides": [1, 2, 4],
                                  "num_upsample_filter": [128, 128, 128]},
            "shrink_header": {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [256], "input_dim": 384}}


def synthetic_pillars(n_agents: int, n_per_agent: int, nx: int, ny: int, args: dict, seed: int = 3):
    import numpy as np
    rs = np.random.RandomState(seed)
    feats, coords, counts = [], [], []
    vx, vy, vz = args["voxel_size"]
    x0, y0, z0 = args["lidar_range"][:3]
    for a in range(n_agents):
        cells = rs.choice(nx * ny, size=n_per_agent, replace=False)
        cy, cx = cells // nx, cells % nx
        n_pts = rs.randint(1, 33, size=n_per_agent)
        pts = np.zeros((n_per_agent, 32, 4), np.float32)
        for i in range(n_per_agent):
            k = n_pts[i]
            pts[i, :k, 0] = x0 + (cx[i] + rs.uniform(0, 1, k)) * vx
            pts[i, :k, 1] = y0 + (cy[i] + rs.uniform(0, 1, k)) * vy
            pts[i, :k, 2] = z0 + rs.uniform(0, 1, k) * vz
            pts[i, :k, 3] = rs.uniform(0, 1, k)
        feats.append(pts)
        coords.append(np.stack([np.full(n_per_agent, a), np.zeros(n_per_agent, np.int64), cy, cx], 1))
        counts.append(n_pts)
    return (torch.from_numpy(np.concatenate(feats)), torch.from_numpy(np.concatenate(coords)).int(),
            torch.from_numpy(np.concatenate(counts)).int())


def random_state_dict(args: dict, seed: int = 0) -> Dict[str, Tensor]:
    """Reference-named PointPillar weights (point_pillar.py:10-33) with non-trivial BN statistics."""
    import math
    import numpy as np
    rs = np.random.RandomState(seed)
    sd: Dict[str, Tensor] = {}
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))

    def conv(name, co, ci, k, bias):
        b = 1.0 / math.sqrt(ci * k * k)
        sd[f"{name}.weight"] = t(rs.uniform(-b, b, (co, ci, k, k)))
        if bias:
            sd[f"{name}.bias"] = t(rs.uniform(-b, b, co))

    def bn(name, c):
        sd[f"{name}.weight"] = t(1 + 0.2 * rs.standard_normal(c))
        sd[f"{name}.bias"] = t(0.2 * rs.standard_normal(c))
        sd[f"{name}.running_mean"] = t(0.3 * rs.standard_normal(c))
        sd[f"{name}.running_var"] = t(rs.uniform(0.5, 1.5, c))
        sd[f"{name}.num_batches_tracked"] = torch.tensor(1)

    sd["pillar_vfe.pfn_layers.0.linear.weight"] = t(rs.uniform(-0.3, 0.3, (64, 10)))
    bn("pillar_vfe.pfn_layers.0.norm", 64)
    bb = args["base_bev_backbone"]
    cin = 64
    for i, (n_layers, co) in enumerate(zip(bb["layer_nums"], bb["num_filters"])):
        conv(f"backbone.blocks.{i}.1", co, cin, 3, False)
        bn(f"backbone.blocks.{i}.2", co)
        for k in range(n_layers):
            conv(f"backbone.blocks.{i}.{4 + 3 * k}", co, co, 3, False)
            bn(f"backbone.blocks.{i}.{5 + 3 * k}", co)
        us, cu = bb["upsample_strides"][i], bb["num_upsample_filter"][i]
        b = 1.0 / math.sqrt(co * us * us)
        sd[f"backbone.deblocks.{i}.0.weight"] = t(rs.uniform(-b, b, (co, cu, us, us)))   # ConvTranspose2d: (in, out, k, k)
        bn(f"backbone.deblocks.{i}.1", cu)
        cin = co
    sh = args["shrink_header"]
    conv("shrink_conv.layers.0.double_conv.0", sh["dim"][0], sh["input_dim"], 3, True)
    conv("shrink_conv.layers.0.double_conv.2", sh["dim"][0], sh["dim"][0], 3, True)
    conv("cls_head", args["anchor_number"], args["cls_head_dim"], 1, True)
    conv("reg_head", 7 * args["anchor_number"], args["cls_head_dim"], 1, True)
    return sd
