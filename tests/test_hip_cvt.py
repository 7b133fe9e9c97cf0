"""HIP camera -> BEV lift (hm-vit_amd/cvt.py, csrc/cvt.hip) against the golden run of the reference's CrossViewAttention
(tests/golden/g11_cross_view.npz) and the oracle at the shipped level-2 size."""
import pytest
import torch

from conftest import load_golden, rel_max_err
from oracle import cvt_oracle as CO

pytestmark = pytest.mark.gpu
TOL = 1e-4   # f32 kernels


def _net(feat_hw, feat_dim, dim, cfg, sd):
    from hmvit_amd.cvt import CrossViewAttention
    net = CrossViewAttention(feat_hw, feat_hw, feat_dim, dim, cfg)
    if cfg["no_image_features"]:
        sd = {k: v for k, v in sd.items() if not k.startswith("feature_proj.")}     # the module has no such branch then
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    return net.cuda().eval()


@pytest.mark.parametrize("tag,no_feat,skip", [("a", False, True), ("b", True, False)])
def test_cross_view_attention_golden(tag, no_feat, skip):
    from hmvit_amd.cvt import BEVEmbedding
    g = load_golden("g11_cross_view.npz")
    cfg = CO.make_config()
    cfg["no_image_features"], cfg["skip"] = no_feat, skip
    sd = CO.random_state_dict(64, 128, cfg, seed=int(g["seed_weights"]))
    net = _net(12, 64, 128, cfg, sd)
    bev = BEVEmbedding(128, 1.0, 64, 64, 100.0, 100.0, 0.0, [128, 128, 64]).cuda()
    assert torch.allclose(bev.grid.cpu(), CO.bev_grid(64, 64, 100.0, 100.0, 0.0, 3))
    x, feat, I_inv, E_inv = CO.synthetic_inputs(2, 4, 64, 12, 12, 128, 8, 8, seed=int(g["seed_inputs"]))
    y = net(x.cuda(), bev, feat.cuda(), I_inv.cuda(), E_inv.cuda()).cpu()
    assert y.shape == g["y_" + tag].shape
    assert rel_max_err(y, g["y_" + tag]) < TOL


def test_cross_view_attention_level2_size_vs_oracle():
    """Second pyramid level of the shipped CVT config: (4, 512, 16, 16) features, 32 x 32 BEV queries, dim 128, 4 heads."""
    from hmvit_amd.cvt import BEVEmbedding
    cfg = CO.make_config()
    sd = CO.random_state_dict(512, 128, cfg, seed=5)
    net = _net(16, 512, 128, cfg, sd)
    bev = BEVEmbedding(128, 1.0, 256, 256, 100.0, 100.0, 0.0, [128, 128, 64]).cuda()
    x, feat, I_inv, E_inv = CO.synthetic_inputs(1, 4, 512, 16, 16, 128, 32, 32, seed=6)
    ref = CO.cross_view_attention(x, bev.grid.cpu(), feat, I_inv, E_inv, sd, cfg)
    y = net(x.cuda(), bev, feat.cuda(), I_inv.cuda(), E_inv.cuda()).cpu()
    assert rel_max_err(y, ref) < TOL
    with pytest.raises(RuntimeError):
        net(x, bev, feat, I_inv, E_inv)          # CPU tensors: no fallback
