"""Point cloud -> pillars on the GPU: mirror of the reference's ``SpVoxelPreprocessor``
(``opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:14-57``), whose work is done by the third-party
``spconv.utils.Point2VoxelCPU3d`` on the host.  Same constructor dict (the yaml's ``preprocess`` block) and the same
``preprocess(pcd)`` keys (``voxel_features (Nv, max_points, 4)``, ``voxel_coords (Nv, 3) [z, y, x]``, ``voxel_num_points
(Nv)``) in the same deterministic order as spconv's sequential algorithm; the values are CUDA tensors instead of numpy
arrays, so the LiDAR encoder can consume them without a host round trip.  ``collate_batch`` prepends the agent index
exactly like ``collate_batch_list`` (:83-120)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


class SpVoxelPreprocessor:
    def __init__(self, preprocess_params: dict, train: bool):
        self.params = preprocess_params
        self.train = train
        self.lidar_range = list(self.params["cav_lidar_range"])
        self.voxel_size = list(self.params["args"]["voxel_size"])
        self.max_points_per_voxel = int(self.params["args"]["max_points_per_voxel"])
        self.max_voxels = int(self.params["args"]["max_voxel_train" if train else "max_voxel_test"])
        grid = (np.array(self.lidar_range[3:6]) - np.array(self.lidar_range[0:3])) / np.array(self.voxel_size)
        self.grid_size = np.round(grid).astype(np.int64)

    def preprocess(self, pcd) -> dict:
        pts = torch.as_tensor(pcd)
        if not pts.is_cuda:
            pts = pts.cuda()              # the reference hands over a host array; the kernels need it on the device
        pts = pts[:, :4].contiguous().float()
        n = pts.shape[0]
        dev = pts.device
        nx, ny, nz = (int(v) for v in self.grid_size)
        ws_bytes = int(_lib.lib.hmvit_voxelize_workspace_bytes(n, nx, ny, nz))
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        voxels = torch.empty(self.max_voxels, self.max_points_per_voxel, 4, device=dev)
        coords = torch.empty(self.max_voxels, 3, device=dev, dtype=torch.int32)
        num = torch.empty(self.max_voxels, device=dev, dtype=torch.int32)
        n_vox = torch.zeros(1, device=dev, dtype=torch.int32)
        vs = (ctypes.c_float * 3)(*self.voxel_size)
        rng = (ctypes.c_float * 6)(*self.lidar_range)
        _lib.check(_lib.lib.hmvit_voxelize(pts.data_ptr(), n, vs, rng, self.max_points_per_voxel, self.max_voxels, ws.data_ptr(),
                                           ws_bytes, voxels.data_ptr(), coords.data_ptr(), num.data_ptr(), n_vox.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "voxelize")
        k = int(n_vox.item())
        return {"voxel_features": voxels[:k], "voxel_coords": coords[:k], "voxel_num_points": num[:k]}

    def collate_batch(self, batch):
        """sp_voxel_preprocessor.py:59-120: concatenate the agents' pillars, coordinates get the agent index prepended."""
        if isinstance(batch, dict):
            batch = [{k: batch[k][i] for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
                     for i in range(len(batch["voxel_features"]))]
        feats = torch.cat([torch.as_tensor(b["voxel_features"]) for b in batch])
        nums = torch.cat([torch.as_tensor(b["voxel_num_points"]) for b in batch])
        coords = torch.cat([torch.nn.functional.pad(torch.as_tensor(b["voxel_coords"]), (1, 0), value=i)
                            for i, b in enumerate(batch)])
        return {"voxel_features": feats, "voxel_coords": coords, "voxel_num_points": nums}
