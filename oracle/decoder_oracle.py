"""CPU oracle for the detection tail (TEST INFRASTRUCTURE): HeteroDecoder + NaiveDecoder in eval mode
(opencood/models/sub_modules/hetero_decoder.py:42-89, naive_decoder.py:63-92, use_upsample=False as on
the HM-ViT path, bevformer_point_pillar_hetero.py:125).  Parity pinned by tests/golden/g8_decoder.npz."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def naive_decoder(x: Tensor, sd: Dict[str, Tensor], prefix: str, num_layer: int) -> Tensor:
    """x (N, C, H, W): per layer conv3x3+BN(eps 1e-5)+ReLU twice; module list index 6*step + {0,1,3,4}."""
    for step in range(num_layer):
        for sub in (0, 3):
            c, b = f"{prefix}.decoder.{6 * step + sub}", f"{prefix}.decoder.{6 * step + sub + 1}"
            x = F.conv2d(x, sd[f"{c}.weight"], sd[f"{c}.bias"], 1, 1)
            x = F.relu(F.batch_norm(x, sd[f"{b}.running_mean"], sd[f"{b}.running_var"], sd[f"{b}.weight"],
                                    sd[f"{b}.bias"], False, 0.0, 1e-5))
    return x


def hetero_decoder(x: Tensor, mode: Tensor, sd: Dict[str, Tensor], params: dict, prefix: str = "",
                   dtype: torch.dtype = torch.float32):
    """x (B, 1, C, H, W), mode (B, L): decoder and heads picked by the ego type mode[:, 0].  dtype=torch.float64: the same
    layers in double precision (yardstick of the precision stress tests)."""
    pre = f"{prefix}." if prefix else ""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    B = x.shape[0]
    psm, rm = [None] * B, [None] * B
    for t, name in ((0, "camera"), (1, "lidar")):
        idx = [b for b in range(B) if int(mode[b, 0]) == t]
        if not idx:
            continue
        f = naive_decoder(x[idx, 0].to(dtype), sd, f"{pre}{name}_decoder", params["num_layer"])
        p = F.conv2d(f, sd[f"{pre}{name}_cls_head.weight"], sd[f"{pre}{name}_cls_head.bias"])
        r = F.conv2d(f, sd[f"{pre}{name}_reg_head.weight"], sd[f"{pre}{name}_reg_head.bias"])
        for j, b in enumerate(idx):
            psm[b], rm[b] = p[j], r[j]
    for b in range(B):
        if psm[b] is None:
            raise ValueError(f"Mode but be either 1 or 0 but received {int(mode[b, 0])}")
    return torch.stack(psm), torch.stack(rm)


def naive_compressor(x: Tensor, sd: Dict[str, Tensor], prefix: str = "") -> Tensor:
    """NaiveCompressor.forward in eval mode (naive_compress.py:5-28): three conv3x3 + BatchNorm(eps 1e-3) + ReLU."""
    pre = f"{prefix}." if prefix else ""
    for c, b in (("encoder.0", "encoder.1"), ("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
        x = F.conv2d(x, sd[f"{pre}{c}.weight"], sd[f"{pre}{c}.bias"], 1, 1)
        x = F.relu(F.batch_norm(x, sd[f"{pre}{b}.running_mean"], sd[f"{pre}{b}.running_var"], sd[f"{pre}{b}.weight"],
                                sd[f"{pre}{b}.bias"], False, 0.0, 1e-3))
    return x


def compressor_state_dict(input_dim: int, ratio: int, seed: int = 0) -> Dict[str, Tensor]:
    import numpy as np
    rs = np.random.RandomState(seed)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    sd: Dict[str, Tensor] = {}
    mid = input_dim // ratio
    for c, b, co, ci in (("encoder.0", "encoder.1", mid, input_dim), ("decoder.0", "decoder.1", input_dim, mid),
                         ("decoder.3", "decoder.4", input_dim, input_dim)):
        bd = 1.0 / math.sqrt(ci * 9)
        sd[f"{c}.weight"] = t(rs.uniform(-bd, bd, (co, ci, 3, 3)))
        sd[f"{c}.bias"] = t(rs.uniform(-bd, bd, co))
        sd[f"{b}.weight"] = t(1 + 0.2 * rs.standard_normal(co))
        sd[f"{b}.bias"] = t(0.2 * rs.standard_normal(co))
        sd[f"{b}.running_mean"] = t(0.3 * rs.standard_normal(co))
        sd[f"{b}.running_var"] = t(rs.uniform(0.5, 1.5, co))
        sd[f"{b}.num_batches_tracked"] = torch.tensor(1)
    return sd


def make_params(input_dim: int = 256, anchor_number: int = 2) -> dict:
    return {"input_dim": input_dim, "num_layer": 2, "num_ch_dec": [256, 256], "anchor_number": anchor_number}


def random_state_dict(params: dict, seed: int = 0, prefix: str = "") -> Dict[str, Tensor]:
    import numpy as np
    rs = np.random.RandomState(seed)
    pre = f"{prefix}." if prefix else ""
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    sd: Dict[str, Tensor] = {}

    def conv(name, co, ci, k):
        b = 1.0 / math.sqrt(ci * k * k)
        sd[f"{name}.weight"] = t(rs.uniform(-b, b, (co, ci, k, k)))
        sd[f"{name}.bias"] = t(rs.uniform(-b, b, co))

    def bn(name, c):
        sd[f"{name}.weight"] = t(1 + 0.2 * rs.standard_normal(c))
        sd[f"{name}.bias"] = t(0.2 * rs.standard_normal(c))
        sd[f"{name}.running_mean"] = t(0.3 * rs.standard_normal(c))
        sd[f"{name}.running_var"] = t(rs.uniform(0.5, 1.5, c))
        sd[f"{name}.num_batches_tracked"] = torch.tensor(1)

    nl, chs = params["num_layer"], params["num_ch_dec"]
    for name in ("camera", "lidar"):
        step = 0
        for i in range(nl - 1, -1, -1):
            cin = params["input_dim"] if i == nl - 1 else chs[i + 1]
            conv(f"{pre}{name}_decoder.decoder.{6 * step}", chs[i], cin, 3)
            bn(f"{pre}{name}_decoder.decoder.{6 * step + 1}", chs[i])
            conv(f"{pre}{name}_decoder.decoder.{6 * step + 3}", chs[i], chs[i], 3)
            bn(f"{pre}{name}_decoder.decoder.{6 * step + 4}", chs[i])
            step += 1
        conv(f"{pre}{name}_cls_head", params["anchor_number"], chs[0], 1)
        conv(f"{pre}{name}_reg_head", 7 * params["anchor_number"], chs[0], 1)
    return sd
