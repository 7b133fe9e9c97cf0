"""Minimal train harness: the loop of the reference's ``opencood/tools/train_camera.py:43-230`` over the
synthetic replay scenes, one process per GPU with the gradient all-reduce on RCCL.

What it mirrors, piece by piece (citations into /root/reference/opencood):
  * process group      ``tools/multi_gpu_utils.py:16-37``   RANK / WORLD_SIZE / LOCAL_RANK from the launcher, backend "nccl"
                                                             (= RCCL over xGMI on ROCm), one device per process;
  * model wrapping     ``tools/train_camera.py:118-131``    ``fix_lidar_backbone()`` then
                                                             ``DistributedDataParallel(model, device_ids=[gpu], find_unused_parameters=True)``
                                                             (the parameters of agent types absent from a batch, ``aggregate_fc`` and the
                                                             unused ``cls_head`` / ``reg_head`` get no gradient);
  * loss / optimiser   ``tools/train_utils.py:146-230``     ``PointPillarLoss`` (hm-vit_amd/train.py), AdamW lr 2e-4 eps 1e-10 wd 1e-2;
  * lr schedule        ``tools/train_utils.py:247-264``     timm's ``CosineLRScheduler`` (third party, absent, version unpinned:
                                                             ``cosine_lr.py`` restated from its published formula, parity unpinned),
                                                             stepped per iteration with ``step_update(epoch * num_steps + i)``;
  * per-batch step     ``tools/train_camera.py:163-199``    ``model.train(); zero_grad(); out = model(batch['ego']);
                                                             loss = criterion(out, label_dict); loss.backward(); optimizer.step()``;
  * checkpoints        ``tools/train_camera.py:221-225``, ``tools/train_utils.py:40-75``: ``net_epoch%d.pth`` = ``state_dict()`` of the
                                                             un-wrapped model every ``save_freq`` epochs; resuming loads the highest epoch
                                                             found in the folder with ``strict=False``;
  * labels             ``data_utils/post_processor/voxel_postprocessor.py:74-229`` (hm-vit_amd/postprocess.py ``generate_label``).

What trains on which code: the fusion (the hot path) runs its HIP forward AND backward kernels (hm-vit_amd/train.py); the LiDAR
encoder runs frozen on its HIP inference kernels (``fix_lidar_backbone``, the reference's own option, the default here) or trains
on libhmvit too (``--train_lidar_backbone``, hm-vit_amd/encoder_train.py);
the detection tail (``HeteroDecoder``: four 3x3 convolutions + BatchNorm in batch-statistics mode + ReLU + two 1x1 heads) runs
HIP forward and backward kernels as well (hm-vit_amd/tail_train.py).

    python -m hmvit_amd.trainer --epochs 2 --frames 8                      (one GPU)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m hmvit_amd.trainer --epochs 2
"""
from __future__ import annotations

import argparse
import glob
import json
import math
import os
import re
import time

import numpy as np
import torch

from . import replay as R
from .train import PointPillarLoss, make_optimizer


# ---------------------------------------------------------------------------------------------------------------------
# schedule, optimiser, loss, checkpoints: pure host logic (tests/test_train_cpu.py)
# ---------------------------------------------------------------------------------------------------------------------
class CosineLRScheduler:
    """timm ``CosineLRScheduler(optimizer, t_initial, lr_min, warmup_lr_init, warmup_t, cycle_limit=1, t_in_epochs=False)`` as
    ``setup_lr_schedular`` constructs it: for update t < warmup_t the rate moves linearly from ``warmup_lr_init`` to the base
    rate, afterwards ``lr_min + (base - lr_min) (1 + cos(pi t / t_initial)) / 2`` (t NOT shifted by the warm-up: timm's
    ``warmup_prefix=False``), and ``lr_min`` once t >= t_initial.  The shipped yaml warms DOWN (warmup_lr 2e-3 > lr 2e-4)."""

    def __init__(self, optimizer, t_initial: int, lr_min: float = 0.0, warmup_lr_init: float = 0.0, warmup_t: int = 0):
        self.optimizer, self.t_initial, self.lr_min = optimizer, max(1, int(t_initial)), lr_min
        self.warmup_lr_init, self.warmup_t = warmup_lr_init, int(warmup_t)
        self.base = [g["lr"] for g in optimizer.param_groups]
        if self.warmup_t:
            for g in optimizer.param_groups:
                g["lr"] = warmup_lr_init

    def lr_at(self, t: int):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (b - self.warmup_lr_init) / self.warmup_t for b in self.base]
        if t >= self.t_initial:
            return [self.lr_min for _ in self.base]
        return [self.lr_min + 0.5 * (b - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for b in self.base]

    def step_update(self, num_updates: int):
        for g, lr in zip(self.optimizer.param_groups, self.lr_at(num_updates)):
            g["lr"] = lr

    def step(self, epoch: int):        # timm steps per epoch only when t_in_epochs; the reference uses step_update
        pass


def default_hypes(epoches: int = 2) -> dict:
    """The train-related blocks of opcl/bevformer_point_pillar_hetero.yaml (:11-20, 156-176)."""
    return {"train_params": {"batch_size": 1, "epoches": epoches, "eval_freq": 1, "save_freq": 1, "max_cav": 5},
            "loss": {"core_method": "point_pillar_loss", "args": {"cls_weight": 1.0, "reg": 2.0}},
            "optimizer": {"core_method": "AdamW", "lr": 2e-4, "args": {"eps": 1e-10, "weight_decay": 1e-2}},
            "lr_scheduler": {"core_method": "cosineannealwarm", "epoches": epoches, "warmup_lr": 2e-3, "warmup_epoches": 10,
                             "lr_min": 5e-6}}


def create_loss(hypes: dict):
    if hypes["loss"]["core_method"] != "point_pillar_loss":
        raise NotImplementedError(f"loss {hypes['loss']['core_method']}: only point_pillar_loss is built")
    return PointPillarLoss(hypes["loss"]["args"])


def setup_optimizer(hypes: dict, model):
    if hypes["optimizer"]["core_method"] != "AdamW":
        raise NotImplementedError("optimizer: the shipped yaml's AdamW")
    return make_optimizer(model.parameters(), hypes["optimizer"])


def setup_lr_schedular(hypes: dict, optimizer, n_iter_per_epoch: int):
    c = hypes["lr_scheduler"]
    if c["core_method"] == "cosineannealwarm":
        return CosineLRScheduler(optimizer, t_initial=c["epoches"] * n_iter_per_epoch, lr_min=c["lr_min"],
                                 warmup_lr_init=c["warmup_lr"], warmup_t=c["warmup_epoches"] * n_iter_per_epoch)
    raise NotImplementedError(f"lr_scheduler {c['core_method']}: only cosineannealwarm (the shipped yaml) is built")


def save_checkpoint(model, saved_path: str, epoch: int) -> str:
    """``torch.save(model_without_ddp.state_dict(), 'net_epoch%d.pth' % (epoch + 1))``."""
    os.makedirs(saved_path, exist_ok=True)
    path = os.path.join(saved_path, "net_epoch%d.pth" % (epoch + 1))
    torch.save(getattr(model, "module", model).state_dict(), path)
    return path


def load_saved_model(saved_path: str, model):
    """train_utils.py:40-75: the highest ``net_epoch%d.pth`` in the folder, ``strict=False``; returns (epoch, model)."""
    if not os.path.exists(saved_path):
        raise FileNotFoundError("{} not found".format(saved_path))
    epochs = [int(m.group(1)) for f in glob.glob(os.path.join(saved_path, "*epoch*.pth"))
              if (m := re.search(r"epoch(\d+)\.pth$", f))]
    initial = max(epochs) if epochs else 0
    if initial > 0:
        state = torch.load(os.path.join(saved_path, "net_epoch%d.pth" % initial), map_location="cpu")
        model.load_state_dict(state, strict=False)
    return initial, model


def init_distributed_mode(backend: str = "nccl") -> dict:
    """multi_gpu_utils.py:16-37 for a torchrun launch on one node; rendezvous on 127.0.0.1 unless the launcher says otherwise.
    ``backend="gloo"`` with more ranks than GPUs (ranks then share devices) is the 1-GPU rehearsal of the loop; "nccl" is RCCL."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return {"distributed": False, "rank": 0, "world_size": 1, "gpu": 0}
    import torch.distributed as dist
    rank, world, gpu = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    gpu %= max(1, torch.cuda.device_count())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(gpu)
    dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    dist.barrier()
    return {"distributed": True, "rank": rank, "world_size": world, "gpu": gpu}


# ---------------------------------------------------------------------------------------------------------------------
# data: the replay scenes with anchor targets
# ---------------------------------------------------------------------------------------------------------------------
class SyntheticTrainDataset(R.SyntheticReplayDataset):
    """``ds[i]`` of the replay dataset plus what ``collate_batch_train`` adds for training
    (``mixed/intermediate_fusion_dataset.py:398-415``): ``object_bbx_center`` / ``object_bbx_mask`` padded to ``max_num`` and the
    ``label_dict`` of ``VoxelPostprocessor.generate_label`` on the ego's anchors."""

    def __init__(self, cfg, n_frames, post, n_agents=None, n_obj=12, seed=7, max_num=100, camera_ratio: float = 0.0,
                 ego_mode: str = "mixed", camera_image: int = 64):
        """``camera_ratio`` / ``ego_mode``: the modality roll of ``basedataset.py:193-200`` - every non-ego agent is a camera agent
        with probability ``camera_to_lidar_ratio``, the ego is ``lidar`` / ``camera`` or rolled like the others (``mixed``).  A frame
        with camera agents carries ``camera`` / ``intrinsic`` / ``extrinsic`` / ``cav2cam_extrinsic`` for ALL its agents next to
        the point clouds, as ``collate_batch`` hands them over (``mixed/intermediate_fusion_dataset.py:398-415``); ``mode`` says
        which sensor of an agent the model reads."""
        super().__init__(cfg, n_frames, n_agents=n_agents, n_obj=n_obj, seed=seed)
        self.post, self.max_num = post, max_num
        self.anchors = post.generate_anchor_box()
        self.camera_ratio, self.ego_mode, self.camera_image = camera_ratio, ego_mode, camera_image

    def roll_modes(self, idx):
        rs = np.random.RandomState(self.seed + 77 + 1000 * idx)
        m = [0 if rs.uniform() < self.camera_ratio else 1 for _ in range(self.n_agents)]
        if self.ego_mode == "lidar":
            m[0] = 1
        elif self.ego_mode == "camera":
            m[0] = 0
        return m

    def __getitem__(self, idx):
        frame = super().__getitem__(idx)
        if self.camera_ratio > 0 or self.ego_mode == "camera":
            from . import synthetic as S
            modes = self.roll_modes(idx)
            mode = torch.zeros(1, self.cfg["max_cav"], dtype=torch.float64)       # padding slots: 0 (base_camera_lidar_dataset.py:178)
            mode[0, : self.n_agents] = torch.tensor(modes, dtype=torch.float64)
            frame["mode"] = mode
            frame.update(S.synthetic_cameras(self.n_agents, self.camera_image, seed=self.seed + 31 + 1000 * idx))
        boxes = frame["object_bbx_center_valid"]
        center = np.zeros((self.max_num, 7), np.float32)
        mask = np.zeros(self.max_num, np.float32)
        center[: len(boxes)], mask[: len(boxes)] = boxes, 1
        frame["object_bbx_center"], frame["object_bbx_mask"] = center, mask
        frame["label_dict"] = self.post.collate_batch([self.post.generate_label(gt_box_center=center, anchors=self.anchors,
                                                                                mask=mask)])
        return frame


def to_batch(frame: dict, pre, device) -> dict:
    """One frame -> the ``batch['ego']`` dict the model and the criterion take (pillarisation on the device, csrc/vox.hip)."""
    lidar = pre.collate_batch([pre.preprocess(c) for c in frame["clouds"]])
    label = {k: v.to(device=device, dtype=torch.float32) for k, v in frame["label_dict"].items()}
    batch = {"mode": frame["mode"].to(device), "record_len": frame["record_len"].to(device),
             "pairwise_t_matrix": frame["pairwise_t_matrix"].to(device), "processed_lidar": lidar, "label_dict": label}
    for k in ("camera", "intrinsic", "extrinsic", "cav2cam_extrinsic"):
        if k in frame:
            batch[k] = frame[k].to(device)
    return batch


# ---------------------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------------------
def validate(model, val_dataset, pre, criterion, device) -> float:
    """The validation pass of ``train_camera.py:201-220``: ``model.eval()`` per batch, ``torch.no_grad()``, mean of the
    criterion over the validation frames (every rank runs the whole split, as the reference's un-sharded ``val_loader``).
    ``eval()`` flips the swapped-in modules to their inference kernels (BatchNorm running statistics, no dropout, the fused
    fusion launch) and ``train()`` at the top of the next step flips them back."""
    losses = []
    with torch.no_grad():
        for i in range(len(val_dataset)):
            model.eval()
            batch = to_batch(val_dataset[i], pre, device)
            losses.append(float(criterion(model(batch), batch["label_dict"])))
    return sum(losses) / max(1, len(losses))


def train(model, dataset, pre, hypes: dict, saved_path: str | None = None, init_epoch: int = 0, dist_info: dict | None = None,
          log=None, val_dataset=None) -> dict:
    """train_camera.py:133-230 (no tensorboard or AMP).  Frames shard over ranks as ``DistributedSampler`` does (rank r takes
    frames r, r + world, ...; the permutation is reseeded per epoch by ``set_epoch``); ``val_dataset``: the validation split,
    run every ``eval_freq`` epochs (:201-220)."""
    info = dist_info or {"distributed": False, "rank": 0, "world_size": 1, "gpu": 0}
    device = next(model.parameters()).device
    model_without_ddp = model
    if info["distributed"]:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[info["gpu"]], find_unused_parameters=True)
        model_without_ddp = model.module
    criterion = create_loss(hypes)
    optimizer = setup_optimizer(hypes, model_without_ddp)
    world, rank = info["world_size"], info["rank"]
    num_steps = (len(dataset) + world - 1) // world
    scheduler = setup_lr_schedular(hypes, optimizer, num_steps)
    epoches = hypes["train_params"]["epoches"]
    history, val_history, t_step, n_step = [], [], 0.0, 0
    for epoch in range(init_epoch, max(epoches, init_epoch)):
        order = np.random.RandomState(epoch).permutation(len(dataset)) if world > 1 else np.arange(len(dataset))
        mine = [int(order[(rank + k * world) % len(order)]) for k in range(num_steps)]      # padded like DistributedSampler
        losses = []
        for i, idx in enumerate(mine):
            batch = to_batch(dataset[idx], pre, device)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            model.train()
            model.zero_grad()
            optimizer.zero_grad()
            out = model(batch)
            loss = criterion(out, batch["label_dict"])
            loss.backward()
            optimizer.step()
            scheduler.step_update(epoch * num_steps + i)
            torch.cuda.synchronize(device)
            if epoch > init_epoch or i > 0:               # the first step carries the weight preparation
                t_step += time.perf_counter() - t0
                n_step += 1
            losses.append(float(loss.detach()))
            if log:
                log(f"[epoch {epoch}][{i + 1}/{num_steps}] loss {losses[-1]:.4f} conf {float(criterion.loss_dict['conf_loss']):.4f} "
                    f"loc {float(criterion.loss_dict['reg_loss']):.4f} lr {optimizer.param_groups[0]['lr']:.2e}")
        history.append(sum(losses) / len(losses))
        if val_dataset is not None and epoch % hypes["train_params"]["eval_freq"] == 0:
            val = validate(model, val_dataset, pre, criterion, device)
            val_history.append(val)
            if log:
                log("At epoch %d, the validation loss is %f" % (epoch, val))
        if saved_path and rank == 0 and epoch % hypes["train_params"]["save_freq"] == 0:
            save_checkpoint(model_without_ddp, saved_path, epoch)
    res = {"epoch_loss": history, "val_loss": val_history, "ms_per_step": 1e3 * t_step / max(1, n_step), "steps": n_step,
           "world_size": world}
    if info["distributed"]:
        # after the last step every rank must hold the same parameters (DDP averaged every gradient): largest difference between
        # the ranks' trainable parameters, relative to their largest magnitude - 0 when the all-reduce did its job
        import torch.distributed as dist
        flat = torch.cat([p.detach().float().reshape(-1) for p in model_without_ddp.parameters() if p.requires_grad])
        hi, lo = flat.clone(), flat.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        res["rank_param_spread"] = float((hi - lo).abs().max() / flat.abs().max().clamp_min(1e-30))
    return res


def build(args):
    from . import BevformerPointPillarHetero, SpVoxelPreprocessor, VoxelPostprocessor
    cfg = R.lidar_model_config(args.grid[0], args.grid[1], max_cav=args.agents, small=args.small)
    torch.manual_seed(args.seed)
    camera_encoder = None
    mixed = args.camera_ratio > 0 or args.ego_mode == "camera"
    if mixed:
        # camera agents: the CVT lift in the model's camera slot, producing the LiDAR branch's (256, ny / 4, nx / 4) BEV map; frozen
        # (train_camera.py's --fix_camera_backbone) unless --train_camera_backbone puts it on the tape (hm-vit_amd/camera_train.py)
        from . import synthetic as S
        from .camera import CvtCameraEncoder
        camera_encoder = CvtCameraEncoder(S.camera_config(args.camera_image, 18, bev_h=args.grid[1] // 2, bev_w=args.grid[0] // 2),
                                          precision="f32" if args.precision == "f32" else "f16")
    model = BevformerPointPillarHetero(cfg, camera_encoder=camera_encoder, precision=args.precision)
    if mixed and not getattr(args, "train_camera_backbone", False):
        model.fix_camera_backbone()                     # train_camera.py --fix_camera_backbone (the default here: faster steps)
    if not args.train_lidar_backbone:
        model.fix_lidar_backbone()
    pre = SpVoxelPreprocessor(R.preprocess_params(cfg), train=True)
    post = VoxelPostprocessor(R.postprocess_params(cfg), train=True)
    kw = dict(n_agents=args.agents, camera_ratio=args.camera_ratio, ego_mode=args.ego_mode, camera_image=args.camera_image)
    ds = SyntheticTrainDataset(cfg, args.frames, post, seed=args.seed + 7, **kw)
    val = SyntheticTrainDataset(cfg, args.val_frames, post, seed=args.seed + 100007, **kw) if args.val_frames > 0 else None
    return cfg, model, pre, post, ds, val


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--grid", type=int, nargs=2, default=[256, 128], metavar=("NX", "NY"))
    ap.add_argument("--small", action="store_true", help="PointPillar layer_nums [1, 2, 2]")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16"], help="precision of the frozen encoder's kernels")
    ap.add_argument("--val_frames", type=int, default=2, help="frames of the validation split (0: no validation pass)")
    ap.add_argument("--camera_ratio", type=float, default=0.0,
                    help="camera_to_lidar_ratio of the modality roll (basedataset.py:193-200); > 0 puts the CVT lift in the camera slot")
    ap.add_argument("--ego_mode", default="mixed", choices=["mixed", "lidar", "camera"])
    ap.add_argument("--camera_image", type=int, default=64, help="camera image size of the synthetic frames")
    ap.add_argument("--train_camera_backbone", action="store_true",
                    help="do NOT freeze the camera encoder: ResNet, cross-view lift and decoder train too (hm-vit_amd/camera_train.py)")
    ap.add_argument("--train_lidar_backbone", action="store_true",
                    help="do NOT freeze the LiDAR encoder: PointPillar trains too (hm-vit_amd/encoder_train.py)")
    ap.add_argument("--model_dir", default=None, help="folder with net_epoch%%d.pth to resume from / save into")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("hm-vit_amd has no CPU path: this needs an MI355X")
    info = init_distributed_mode(args.backend)
    hypes = default_hypes(args.epochs)
    cfg, model, pre, post, ds, val = build(args)
    init_epoch = 0
    if args.model_dir and os.path.exists(args.model_dir):
        init_epoch, model = load_saved_model(args.model_dir, model)
    model = model.to(f"cuda:{info['gpu']}")
    res = train(model, ds, pre, hypes, saved_path=args.model_dir, init_epoch=init_epoch, dist_info=info,
                log=print if args.verbose and info["rank"] == 0 else None, val_dataset=val)
    if info["rank"] == 0:
        res.update(agents=args.agents, grid=args.grid, frames=args.frames)
        print(json.dumps(res))
    if info["distributed"]:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
