"""HIP pillariser (hm-vit_amd/voxelizer.py, csrc/vox.hip) against the sequential restatement of spconv's algorithm."""
import numpy as np
import pytest
import torch

from oracle import voxelizer_oracle as VO

pytestmark = pytest.mark.gpu

PARAMS = {"cav_lidar_range": [-12.8, -9.6, -3, 12.8, 9.6, 1],
          "args": {"voxel_size": [0.4, 0.4, 4], "max_points_per_voxel": 32, "max_voxel_train": 32000, "max_voxel_test": 70000}}


def _check(params, cloud, train):
    import hmvit_amd
    from hmvit_amd.voxelizer import SpVoxelPreprocessor
    pp = SpVoxelPreprocessor(params, train=train)
    out = pp.preprocess(cloud)
    rv, rc, rn = VO.point_to_voxel(cloud, params["args"]["voxel_size"], params["cav_lidar_range"],
                                   params["args"]["max_points_per_voxel"], pp.max_voxels)
    assert out["voxel_features"].shape == rv.shape
    assert np.array_equal(out["voxel_coords"].cpu().numpy(), rc)
    assert np.array_equal(out["voxel_num_points"].cpu().numpy(), rn)
    assert np.array_equal(out["voxel_features"].cpu().numpy(), rv)      # bit-exact: values are copied, order is defined
    return out


def test_voxelize_matches_sequential_algorithm():
    cloud = VO.synthetic_cloud(20000, PARAMS["cav_lidar_range"], seed=1)
    out = _check(PARAMS, cloud, train=False)
    assert int(out["voxel_num_points"].max()) == 32          # the clusters overflow cells
    assert out["voxel_features"].shape[0] > 1000


def test_voxelize_voxel_cap_and_edge_cases():
    params = {"cav_lidar_range": PARAMS["cav_lidar_range"],
              "args": dict(PARAMS["args"], max_voxel_train=300, max_points_per_voxel=5)}
    cloud = VO.synthetic_cloud(5000, params["cav_lidar_range"], seed=2)
    out = _check(params, cloud, train=True)                  # 300-voxel cap: later cells are dropped, earlier ones keep filling
    assert out["voxel_features"].shape[0] == 300
    # every point outside the range -> no voxel
    far = cloud.copy(); far[:, 0] += 1000.0
    assert _check(params, far, train=True)["voxel_features"].shape[0] == 0
    # points exactly on the lower / upper range boundary (floor semantics: lower edge inside, upper edge outside)
    edge = np.array([[-12.8, -9.6, -3.0, 0.5], [12.8, 0.0, 0.0, 0.5], [12.79999, 9.59999, 0.99, 0.1]], np.float32)
    _check(PARAMS, edge, train=False)


def test_collate_prepends_agent_index():
    from hmvit_amd.voxelizer import SpVoxelPreprocessor
    pp = SpVoxelPreprocessor(PARAMS, train=False)
    a = pp.preprocess(VO.synthetic_cloud(3000, PARAMS["cav_lidar_range"], seed=3))
    b = pp.preprocess(VO.synthetic_cloud(2000, PARAMS["cav_lidar_range"], seed=4))
    batch = pp.collate_batch([a, b])
    na = a["voxel_coords"].shape[0]
    assert batch["voxel_coords"].shape[1] == 4
    assert int(batch["voxel_coords"][:na, 0].max()) == 0 and int(batch["voxel_coords"][na:, 0].min()) == 1
    assert torch.equal(batch["voxel_coords"][na:, 1:], b["voxel_coords"])


@pytest.mark.parametrize("seed", range(6))
def test_voxelize_random_sweep(seed):
    """Random ranges, voxel sizes and caps, clouds with duplicates and out-of-range points: bit-exact against the sequential
    restatement every time."""
    rs = np.random.RandomState(100 + seed)
    half_x, half_y = float(rs.choice([6.4, 12.8, 25.6])), float(rs.choice([4.8, 9.6, 19.2]))
    params = {"cav_lidar_range": [-half_x, -half_y, -3, half_x, half_y, 1],
              "args": {"voxel_size": [float(rs.choice([0.2, 0.4, 0.8]))] * 2 + [4], "max_points_per_voxel": int(rs.choice([1, 5, 32])),
                       "max_voxel_train": int(rs.choice([50, 700, 32000])), "max_voxel_test": 70000}}
    n = int(rs.choice([1, 37, 4000, 30000]))
    cloud = VO.synthetic_cloud(n, params["cav_lidar_range"], seed=200 + seed)
    cloud[rs.rand(n) < 0.1, 0] += 3 * half_x                       # a tenth of the points outside the range
    if n > 10:
        cloud[n // 2:n // 2 + 5] = cloud[0]                        # exact duplicates of the first point
    _check(params, cloud, train=bool(seed % 2))
