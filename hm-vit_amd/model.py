"""The HM-ViT detector assembled from the native pieces: drop-in for
``opencood/models/bevformer_point_pillar_hetero.py:52-134`` (``BevformerPointPillarHetero``).

Same config dict and ``forward(batch) -> {'psm', 'rm'}`` contract: ``mode`` (B, L),
``record_len`` (B), ``pairwise_t_matrix`` (B, L, L, 4, 4), ``processed_lidar`` {voxel_features,
voxel_coords [agent, z, y, x], voxel_num_points}; same sub-module names (``lidar_encoder``,
``fusion_net``, ``decoder``, ``cls_head``, ``reg_head``) so checkpoints load by key.

The camera slot: the reference wires BEVFormer (mmdet3d, out of scope).  Any module honouring the slot
contract (``set_return_features()``, ``forward(batch_camera) -> (N_cam, 256, H, W)``) can be passed as
``camera_encoder``; without one, a batch that contains camera agents raises ``NotImplementedError``.
The glue between the kernels (modality split, re-interleave, pad to (B, L)) is tensor indexing, as in
``base_camera_lidar_intermediate.py:19-99`` / ``fuse_utils.py:8-61``.
"""
from __future__ import annotations

import torch
from torch import nn

from .decoder import HeteroDecoder, NaiveCompressor
from .fusion import HeteroFusion
from .pointpillar import PointPillar


def regroup(features: torch.Tensor, record_len, max_len: int):
    """(sum N, C, H, W) -> (B, max_len, C, H, W) zero padded + (B, max_len) 0/1 mask (fuse_utils.py:8-61)."""
    lens = [int(v) for v in record_len]
    B = len(lens)
    if all(n == max_len for n in lens):             # nothing to pad: a view
        return features.reshape((B, max_len) + tuple(features.shape[1:])), torch.ones(B, max_len, dtype=torch.int64)
    out = features.new_zeros((B, max_len) + tuple(features.shape[1:]))
    mask = torch.zeros(B, max_len, dtype=torch.int64)
    start = 0
    for b, n in enumerate(lens):
        out[b, :n] = features[start:start + n]
        mask[b, :n] = 1
        start += n
    return out, mask


class BevformerPointPillarHetero(nn.Module):
    def __init__(self, config: dict, camera_encoder: nn.Module = None, precision: str = "split"):
        super().__init__()
        self.camera_encoder = camera_encoder
        fusion_precision = precision
        if precision in ("split", "mixed"):
            # fp32-parity modes: the convolutional encoders / decoder keep f32 maps and run their convolutions on split-f16
            # operands too (csrc/enc.hip k_conv<float, ..., SPLIT>); "mixed" only differs inside the fusion
            precision = "split"
        self.lidar_encoder = PointPillar(config["lidar"], precision=precision)
        self.compression = False
        if config.get("compression", 0) > 0:             # bevformer_point_pillar_hetero.py:69-71 (the shipped yaml uses 0)
            self.compression = True
            self.compressor = NaiveCompressor(256, config["compression"], precision=precision)   # the reference's attribute name
        self.fusion_net = HeteroFusion(config["hetero_fusion"], precision=fusion_precision)
        self.lidar_encoder.set_return_features()
        if self.camera_encoder is not None:
            self.camera_encoder.set_return_features()
        self.use_hetero_decoder = "hetero_decoder" in config
        if not self.use_hetero_decoder:
            raise NotImplementedError("only the hetero_decoder tail of the shipped yaml is built")
        self.decoder = HeteroDecoder(config["hetero_decoder"], precision=precision)
        # present in the reference model regardless of the decoder kind (:74-77); unused with hetero_decoder
        self.cls_head = nn.Conv2d(256, config["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(256, 7 * config["anchor_number"], kernel_size=1)
        self._fix_camera_backbone = False
        self._fix_lidar_backbone = False

    # bevformer_point_pillar_hetero.py:78-89: the train script's --fix_camera_backbone / --fix_lidar_backbone
    def fix_camera_backbone(self):
        self._fix_camera_backbone = True

    def fix_lidar_backbone(self):
        self._fix_lidar_backbone = True

    @staticmethod
    def _freeze_weights(model):
        model.eval()
        for param in model.parameters():
            param.requires_grad = False

    @staticmethod
    def _unpad(mode, record_len):
        return torch.cat([mode[b, :int(n)] for b, n in enumerate(record_len)], dim=0)

    def _lidar_batch(self, batch, flat_mode):
        """extract_lidar_input (base_camera_lidar_intermediate.py:31-65) without mutating the batch:
        pillars of the LiDAR agents, agent index renumbered to the order among LiDAR agents."""
        lid = batch["processed_lidar"]
        if bool((flat_mode == 1).all()):
            # every agent is a LiDAR agent: nothing to filter or renumber (and no boolean-mask indexing, whose output
            # size costs a device synchronisation per forward)
            return {"processed_lidar": {k: lid[k] for k in ("voxel_features", "voxel_coords", "voxel_num_points")},
                    "n_agents": int(flat_mode.numel())}
        coords = lid["voxel_coords"]
        agent = coords[:, 0].long()
        is_lidar = flat_mode.to(coords.device) == 1
        if agent.numel() and int(agent.max()) >= is_lidar.numel():
            raise ValueError("voxel_coords refer to more agents than record_len declares")
        new_index = torch.cumsum(is_lidar.long(), 0) - 1
        keep = is_lidar[agent]
        new_coords = coords[keep].clone()
        new_coords[:, 0] = new_index[agent[keep]].to(coords.dtype)
        return {"processed_lidar": {"voxel_features": lid["voxel_features"][keep], "voxel_coords": new_coords,
                                    "voxel_num_points": lid["voxel_num_points"][keep]},
                "n_agents": int(is_lidar.sum())}

    def forward(self, batch):
        # mode / record_len are needed on the host (regrouping, kernel descriptors): ONE combined read-back per forward (the
        # reference synchronises per agent, bevformer_point_pillar_hetero.py:97-112); no caching by tensor identity -- a
        # loader's fresh tensors reuse addresses.  The host copies are handed down, so the fusion does not read back again.
        if self._fix_lidar_backbone:
            self._freeze_weights(self.lidar_encoder)
        if self._fix_camera_backbone and self.camera_encoder is not None:
            self._freeze_weights(self.camera_encoder)
        m_t, r_t = batch["mode"], batch["record_len"]
        if m_t.device.type != "cpu" or r_t.device.type != "cpu":
            dev = m_t.device if m_t.device.type != "cpu" else r_t.device
            flat = torch.cat([m_t.to(dev).reshape(-1).to(torch.int64), r_t.to(dev).reshape(-1).to(torch.int64)]).cpu()
            mode = flat[:m_t.numel()].reshape(m_t.shape).to(torch.int)
            record_len = flat[m_t.numel():].reshape(r_t.shape)
        else:
            mode, record_len = m_t.to(torch.int), r_t.to(torch.int64)
        rl = [int(v) for v in record_len.tolist()]
        pairwise_t_matrix = batch["pairwise_t_matrix"]
        max_cav = mode.shape[1]
        flat_mode = self._unpad(mode, rl)
        bad = (flat_mode != 0) & (flat_mode != 1)
        if bool(bad.any()):
            raise ValueError(f"Mode but be either 1 or 0 but received {int(flat_mode[bad][0])}")

        camera_features = lidar_features = None
        if not bool((flat_mode == 1).all()):
            if self.camera_encoder is None:
                raise NotImplementedError("this batch has camera agents but no camera_encoder module was supplied "
                                          "(the reference's BEVFormer is out of scope)")
            cam = flat_mode == 0
            batch_camera = {k: batch[k][cam.to(batch[k].device)] for k in
                            ("camera", "intrinsic", "extrinsic", "cav2cam_extrinsic")}
            camera_features = self.camera_encoder(batch_camera)
        if not bool((flat_mode == 0).all()):
            lidar_features = self.lidar_encoder(self._lidar_batch(batch, flat_mode))
        if camera_features is None:
            x = lidar_features                      # single-modality batches need no interleaving (and no masked writes)
        elif lidar_features is None:
            x = camera_features
        else:
            x = lidar_features.new_empty((flat_mode.numel(),) + tuple(lidar_features.shape[1:]))
            x[(flat_mode == 0).to(x.device)] = camera_features.to(x.dtype)
            x[(flat_mode == 1).to(x.device)] = lidar_features
        if self.compression:                              # :116-117, on the concatenated agent maps
            x = self.compressor(x)
        x, mask = regroup(x, rl, max_cav)
        fused = self.fusion_net(x, pairwise_t_matrix, mode, record_len, mask)
        psm, rm = self.decoder(fused.unsqueeze(1), mode, use_upsample=False)
        return {"psm": psm, "rm": rm}
