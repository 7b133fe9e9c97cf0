"""GPU test of the train harness (hm-vit_amd/trainer.py): the reference's per-batch loop on synthetic replay scenes
with the fusion's HIP forward + backward, a frozen HIP LiDAR encoder, checkpoints and resume."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Args:
    grid, agents, small, precision, seed, frames, train_lidar_backbone = [128, 96], 3, True, "f32", 0, 4, False


def test_train_loop_learns_saves_and_resumes(tmp_path):
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T
    hypes = T.default_hypes(epoches=3)
    cfg, model, pre, post, ds = T.build(_Args)
    model = model.cuda()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    res = T.train(model, ds, pre, hypes, saved_path=str(tmp_path))
    assert len(res["epoch_loss"]) == 3 and all(l == l for l in res["epoch_loss"])          # finite
    assert res["epoch_loss"][-1] < res["epoch_loss"][0], res                                  # the loop optimises
    after = model.state_dict()
    moved = lambda k: float((after[k].float() - before[k].float()).abs().max())
    # the fusion (HIP backward) and the tail (torch modules) moved; the frozen encoder and the unused heads did not
    assert moved("fusion_net.hetero_fusion_block.window_attention.q_linears.1.weight") > 0
    assert moved("fusion_net.hetero_fusion_block.window_attention.relation_att") > 0
    assert moved("fusion_net.mlp_head.net.1.0.weight") > 0
    assert moved("decoder.lidar_cls_head.weight") > 0
    assert all(moved(k) == 0 for k in before if k.startswith("lidar_encoder.") and "num_batches" not in k)
    assert moved("fusion_net.hetero_fusion_block.window_attention.q_linears.0.weight") == 0     # camera-type weights: no agent
    assert moved("cls_head.weight") == 0
    assert sorted(os.listdir(tmp_path)) == ["net_epoch1.pth", "net_epoch2.pth", "net_epoch3.pth"]
    # resume: a fresh model picks up epoch 3's weights (train_utils.py:40-75)
    _, fresh, _, _, _ = T.build(_Args)
    epoch, fresh = T.load_saved_model(str(tmp_path), fresh)
    assert epoch == 3
    for k, v in fresh.state_dict().items():
        assert torch.equal(v.cpu(), after[k].cpu()), k
    # evaluation after training runs on the HIP inference kernels (BatchNorm folded) and agrees with the torch modules in eval mode
    model.eval()
    batch = T.to_batch(ds[0], pre, "cuda")
    with torch.no_grad():
        out = model(batch)
        x, mask = hmvit_amd.model.regroup(model.lidar_encoder(model._lidar_batch(batch, torch.ones(3, dtype=torch.int))), [3], 3)
        fused = model.fusion_net(x, batch["pairwise_t_matrix"], batch["mode"].int(), batch["record_len"], mask)
        ref_psm, ref_rm = model.decoder._forward_training(fused.unsqueeze(1), torch.ones(1, 3, dtype=torch.int))
    assert float((out["psm"] - ref_psm).abs().max() / ref_psm.abs().max()) < 1e-4
    assert float((out["rm"] - ref_rm).abs().max() / ref_rm.abs().max()) < 1e-4


def test_unfrozen_encoder_raises():
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T

    class A(_Args):
        train_lidar_backbone = True
    cfg, model, pre, post, ds = T.build(A)
    model = model.cuda().train()
    with pytest.raises(RuntimeError):
        model(T.to_batch(ds[0], pre, "cuda"))
