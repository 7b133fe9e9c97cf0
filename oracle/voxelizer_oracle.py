"""CPU restatement of spconv.utils.Point2VoxelCPU3d.point_to_voxel (the pillariser behind the reference's
SpVoxelPreprocessor, opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:34-57).  TEST INFRASTRUCTURE ONLY.

spconv is a third-party dependency that is neither under /root/reference nor installed here (spconv-cu113, version
unpinned: README.md:27-28), so this follows its published sequential algorithm and the reference's call site; the reference
has no test or fixture for it.  PARITY UNPINNED."""
from __future__ import annotations

import numpy as np


def point_to_voxel(points, voxel_size, lidar_range, max_points, max_voxels):
    pts = np.asarray(points, dtype=np.float32)[:, :4]
    vs = np.asarray(voxel_size, dtype=np.float32)
    lo = np.asarray(lidar_range[:3], dtype=np.float32)
    grid = np.round((np.asarray(lidar_range[3:6]) - np.asarray(lidar_range[:3])) / np.asarray(voxel_size)).astype(np.int64)
    c = np.floor((pts[:, :3] - lo) / vs).astype(np.int64)          # f32 arithmetic, as the C++ loop
    ok = np.all((c >= 0) & (c < grid), axis=1)
    voxels = np.zeros((max_voxels, max_points, 4), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros(max_voxels, np.int32)
    index = {}
    n = 0
    for i in np.nonzero(ok)[0]:
        key = (int(c[i, 2]), int(c[i, 1]), int(c[i, 0]))
        v = index.get(key)
        if v is None:
            if n >= max_voxels:
                continue
            v = n
            index[key] = v
            coords[v] = key
            n += 1
        if num[v] < max_points:
            voxels[v, num[v]] = pts[i]
            num[v] += 1
    return voxels[:n], coords[:n], num[:n]


def synthetic_cloud(n, lidar_range, seed=0, clustered=True):
    """Points mostly inside the range, some outside, with dense clusters so that cells overflow max_points."""
    rs = np.random.RandomState(seed)
    lo, hi = np.asarray(lidar_range[:3], np.float32), np.asarray(lidar_range[3:6], np.float32)
    p = rs.uniform(lo - 2.0, hi + 2.0, size=(n, 3)).astype(np.float32)
    if clustered:
        k = n // 4
        centres = rs.uniform(lo + 1.0, hi - 1.0, size=(12, 3)).astype(np.float32)
        p[:k] = centres[rs.randint(0, 12, k)] + 0.15 * rs.randn(k, 3).astype(np.float32)
        p = p[rs.permutation(n)]
    inten = rs.uniform(0, 1, size=(n, 1)).astype(np.float32)
    return np.concatenate([p, inten], 1)
