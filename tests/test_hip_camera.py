"""HIP camera branch (hm-vit_amd/camera.py) against the CPU restatement (oracle/camera_oracle.py): ResNet pyramid,
cross-view module with bottlenecks, up-sampling decoder, and the assembled encoder."""
import pytest
import torch

from conftest import rel_max_err
from oracle import camera_oracle as CAM

pytestmark = pytest.mark.gpu


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


@pytest.mark.parametrize("precision,tol", [("f32", 2e-4), ("split", 2e-4), ("f16", 1e-2)])
def test_resnet_encoder_vs_oracle(precision, tol):
    from hmvit_amd.camera import ResnetEncoder
    cfg = CAM.make_config(image=64, num_layers=18)
    sd = CAM.random_state_dict(cfg, seed=3)
    net = ResnetEncoder(cfg["encoder"], precision=precision)
    own = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    missing, unexpected = net.load_state_dict(own, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing)
    net = net.cuda().eval()
    batch = CAM.synthetic_batch(2, cfg, seed=4)
    ref = CAM.resnet_encoder(batch["camera"][None], sd, cfg["encoder"], prefix="encoder.encoder")
    out = net(batch["camera"][None].cuda())
    assert [tuple(o.shape) for o in out] == [tuple(r.shape) for r in ref]
    for o, r in zip(out, ref):
        assert o.shape == r.shape and rel_max_err(o.cpu(), r) < tol


@pytest.mark.parametrize("precision,tol", [("f32", 3e-4), ("split", 3e-4), ("f16", 1e-2)])
def test_camera_encoder_vs_oracle(precision, tol):
    from hmvit_amd.camera import CvtCameraEncoder
    cfg = CAM.make_config(image=64, num_layers=18)
    sd = CAM.random_state_dict(cfg, seed=5)
    net = CvtCameraEncoder(cfg, precision=precision)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    net = net.cuda().eval()
    net.set_return_features()
    batch = CAM.synthetic_batch(3, cfg, seed=6)
    ref = CAM.camera_encoder(batch, sd, cfg)
    y = net(_cuda(batch)).cpu()
    assert y.shape == ref.shape == (3, 256, 16, 16)
    assert rel_max_err(y, ref) < tol


def test_camera_encoder_resnet34_shapes():
    from hmvit_amd.camera import ResnetEncoder
    cfg = CAM.make_config(image=128, num_layers=34)
    net = ResnetEncoder(cfg["encoder"], precision="f16").cuda().eval()
    out = net(torch.randn(1, 1, 4, 128, 128, 3).cuda())
    assert [tuple(o.shape) for o in out] == [(1, 1, 4, 128, 16, 16), (1, 1, 4, 512, 4, 4)]
    with pytest.raises(ValueError):
        ResnetEncoder(dict(cfg["encoder"], num_layers=42))


@pytest.mark.parametrize("precision,tol", [("f32", 3e-4), ("split", 3e-4), ("f16", 1e-2)])
def test_resnet50_bottleneck_trunk_vs_oracle(precision, tol):
    """ResNet-50 (torchvision Bottleneck trunk, resnet_ms.py:27-31): pyramid of 256 / 512 / 1024 / 2048 channels."""
    from hmvit_amd.camera import ResnetEncoder
    cfg = CAM.make_config(image=64, num_layers=50)
    cfg["encoder"]["id_pick"] = [0, 1, 2, 3]
    sd = CAM.random_state_dict(cfg, seed=9)
    net = ResnetEncoder(cfg["encoder"], precision=precision)
    own = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    missing, unexpected = net.load_state_dict(own, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing)
    net = net.cuda().eval()
    batch = CAM.synthetic_batch(1, cfg, seed=10)
    ref = CAM.resnet_encoder(batch["camera"][None], sd, cfg["encoder"], prefix="encoder.encoder")
    out = net(batch["camera"][None].cuda())
    assert [o.shape[3] for o in out] == [256, 512, 1024, 2048]
    assert [tuple(o.shape) for o in out] == [tuple(sh[:2]) + (4,) + tuple(sh[3:]) for sh in net.output_shapes]
    for o, r in zip(out, ref):
        assert o.shape == r.shape and rel_max_err(o.cpu(), r) < tol
