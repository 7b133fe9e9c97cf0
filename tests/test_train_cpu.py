"""CPU tests of the training-side host code (hm-vit_amd/train.py): the loss against the reference's own PointPillarLoss
(golden g14, frozen from the imported reference), the differentiable weight folds, and the data-parallel contract
(2 gloo ranks: all-reduced gradients = mean of the per-rank gradients, unused parameters do not stall the reducer)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, load_golden, rel_max_err
from oracle import hmvit_oracle as O

sys.path.insert(0, GOLDEN)


def test_point_pillar_loss_matches_reference_golden():
    from make_goldens_inputs import loss_inputs
    from hmvit_amd.train import PointPillarLoss
    g = load_golden("g14_loss.npz")
    psm, rm, tgt = loss_inputs()
    psm.requires_grad_(True)
    rm.requires_grad_(True)
    crit = PointPillarLoss({"cls_weight": 1.0, "reg": 2.0})
    total = crit({"psm": psm, "rm": rm}, tgt)
    total.backward()
    assert abs(float(total.detach()) - float(g["total"])) < 1e-5 * abs(float(g["total"]))
    assert abs(float(crit.loss_dict["reg_loss"]) - float(g["reg"])) < 1e-5 * abs(float(g["reg"]))
    assert abs(float(crit.loss_dict["conf_loss"]) - float(g["conf"])) < 1e-5 * abs(float(g["conf"]))
    assert rel_max_err(psm.grad, g["d_psm"]) < 1e-5
    assert rel_max_err(rm.grad, g["d_rm"]) < 1e-5


def test_differentiable_folds_equal_the_inference_folds_and_carry_gradients():
    """weights.fold_stage(keep_graph=True) is the algebra the backward kernels' gradients travel back through."""
    from hmvit_amd import weights
    cfg = O.make_config(64, 4, 3)
    sd = O.random_state_dict(cfg, seed=9)
    live = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    a = weights.fold_stage(sd, "hetero_fusion_block", "window", 32, 4, torch.float32)
    b = weights.fold_stage(live, "hetero_fusion_block", "window", 32, 4, torch.float32, keep_graph=True)
    for k in a:
        assert torch.equal(a[k], b[k].detach()), k
    # negated-offset fragments: bias_frag_neg[h, v, lane, r] of tile offset (qt - kt) equals bias_frag of offset (kt - qt)
    # with the roles of row and column swapped -> same multiset of table entries
    assert torch.equal(b["bias_frag_neg"].flatten().sort().values, b["bias_frag"].detach().flatten().sort().values)
    loss = sum((v * torch.arange(v.numel(), dtype=torch.float32).reshape(v.shape) * 1e-3).sum() for k, v in b.items() if v.requires_grad)
    loss.backward()
    p = "hetero_fusion_block.window_attention"
    for name in (f"{p}.relation_att", f"{p}.relation_msg", f"{p}.k_linears.0.weight", f"{p}.v_linears.1.bias",
                 f"{p}.q_linears.1.weight", f"{p}.relative_position_bias_table.weight", "hetero_fusion_block.window_norm.net.0.weight"):
        assert live[name].grad is not None and float(live[name].grad.abs().max()) > 0, name


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _StandIn(torch.nn.Module):
    """A CPU stand-in with the parameter set of HeteroFusion (names, shapes, the never-used aggregate_fc) whose forward is a
    cheap differentiable function of the USED parameters: what DistributedDataParallel sees of the real module."""

    def __init__(self, cfg):
        super().__init__()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        import importlib.util
        spec = importlib.util.spec_from_file_location("_fusion_params", os.path.join(root, "tests", "fusion_params.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        self.net = m.parameter_skeleton(cfg)

    def forward(self, x):
        y = x.sum() * 0
        for n, p in self.net.named_parameters():
            if "aggregate_fc" in n:
                continue
            y = y + (p * x.mean()).sum()
        return y


def _ddp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel
    cfg = O.make_config(64, 4, 2)
    torch.manual_seed(0)
    model = _StandIn(cfg)
    ddp = DistributedDataParallel(model, find_unused_parameters=True)     # train_camera.py:126-131
    x = torch.full((4,), float(rank + 1))                                 # rank-dependent data
    for _ in range(2):                                                     # the reducer must re-arm with unused parameters
        ddp.zero_grad()
        ddp(x).backward()
    g = {n: (p.grad.numpy().copy() if p.grad is not None else None) for n, p in model.net.named_parameters()}
    # per-rank gradients without DDP
    torch.manual_seed(0)
    solo = _StandIn(cfg)
    solo(x).backward()
    own = {n: (p.grad.numpy().copy() if p.grad is not None else None) for n, p in solo.net.named_parameters()}
    gathered = [None] * world
    dist.all_gather_object(gathered, own)
    if rank == 0:
        out.put((g, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_ddp_averages_gradients_and_tolerates_unused_parameters():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g, gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_used = 0
    for name, grad in g.items():
        if "aggregate_fc" in name:
            assert grad is None or float(abs(grad).max()) == 0.0
            continue
        mean = (gathered[0][name] + gathered[1][name]) / 2
        assert torch.allclose(torch.from_numpy(grad), torch.from_numpy(mean), rtol=1e-6, atol=1e-7), name
        n_used += 1
    assert n_used > 40


def test_cosine_schedule_and_checkpoint_naming(tmp_path):
    """Host logic of the train harness (hm-vit_amd/trainer.py): timm's cosine schedule as train_utils.py:247-264 configures it
    and the net_epoch%d.pth checkpoint convention of train_camera.py:221-225 / train_utils.py:40-75."""
    import math
    import hmvit_amd  # noqa: F401
    from hmvit_amd import trainer as T
    lin = torch.nn.Linear(4, 4)
    hypes = T.default_hypes(epoches=40)
    opt = T.setup_optimizer(hypes, lin)
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["eps"] == 1e-10 and opt.defaults["weight_decay"] == 1e-2
    n_iter = 5
    sch = T.setup_lr_schedular(hypes, opt, n_iter)
    assert opt.param_groups[0]["lr"] == 2e-3                               # starts at warmup_lr
    sch.step_update(25)                                                    # half-way through the 10-epoch warm-up
    assert abs(opt.param_groups[0]["lr"] - (2e-3 + 25 * (2e-4 - 2e-3) / 50)) < 1e-12
    sch.step_update(100)                                                   # cosine phase, t not shifted by the warm-up
    want = 5e-6 + 0.5 * (2e-4 - 5e-6) * (1 + math.cos(math.pi * 100 / 200))
    assert abs(opt.param_groups[0]["lr"] - want) < 1e-12
    sch.step_update(200)
    assert opt.param_groups[0]["lr"] == 5e-6
    # checkpoints
    assert T.load_saved_model(str(tmp_path), lin)[0] == 0
    T.save_checkpoint(lin, str(tmp_path), 0)
    with torch.no_grad():
        lin.weight.add_(1.0)
    T.save_checkpoint(lin, str(tmp_path), 6)
    assert sorted(os.listdir(tmp_path)) == ["net_epoch1.pth", "net_epoch7.pth"]
    fresh = torch.nn.Linear(4, 4)
    epoch, fresh = T.load_saved_model(str(tmp_path), fresh)
    assert epoch == 7 and torch.equal(fresh.weight, lin.weight)
    with pytest.raises(NotImplementedError):
        T.setup_lr_schedular({"lr_scheduler": {"core_method": "step"}}, opt, 1)
