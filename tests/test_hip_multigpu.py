"""Multi-GPU runs of the commands the round-end driver uses: they need at least two visible GPUs and skip
on the one-GPU boxes of the build loop, so that a multi-GPU node picks them up by itself.  One process per GPU, RCCL
(`--backend nccl`) over xGMI, rendezvous on 127.0.0.1.  Reference: opencood/tools/train_camera.py:126-131 (DistributedDataParallel,
find_unused_parameters=True), opencood/tools/multi_gpu_utils.py:16-37 (init_distributed_mode)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

N_GPUS = torch.cuda.device_count()
needs_two = pytest.mark.skipif(N_GPUS < 2, reason=f"needs >= 2 GPUs (this box has {N_GPUS})")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    return json.loads(lines[0])


@needs_two
def test_bench_inference_on_all_gpus():
    """`python bench.py --gpus N` (self-spawned ranks, one scene replica per GPU, no data-path collective): one JSON line with
    n_gpus = N and an aggregate above a single replica's."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N_GPUS), "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-strict"], env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _json_line(out)
    assert r["n_gpus"] == N_GPUS and r["scaling"] == "weak" and r["value"] > 0
    assert abs(r["value"] - N_GPUS * 5 / (r["ms_per_step"] * 5e-3)) / r["value"] < 1e-6


@needs_two
def test_bench_inference_under_torch_distributed_run():
    """The driver's launch form: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(N_GPUS), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(N_GPUS), "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-strict"]
    out = subprocess.run(cmd, env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _json_line(out)
    assert r["n_gpus"] == N_GPUS and r["value"] > 0


@needs_two
def test_bench_train_step_with_rccl_allreduce():
    """`python bench.py --train --gpus N --backend nccl`: DistributedDataParallel train step per rank, the gradient all-reduce
    on RCCL the only exchange."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--train", "--gpus", str(N_GPUS), "--steps", "3", "--warmup", "1",
                          "--backend", "nccl", "--config", "native"], env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _json_line(out)
    assert r["n_gpus"] == N_GPUS and r["unit"] == "steps/s" and r["value"] > 0
    assert r["gradient_bytes"] > 0 and r["allreduce_standalone_ms"] > 0 and r["ms_per_step_no_sync"] > 0


@needs_two
def test_trainer_loop_with_rccl_keeps_the_ranks_identical():
    """The train loop (hm-vit_amd/trainer.py) on all GPUs: sharded frames, DDP over RCCL, finite and falling loss, and after the
    last step every rank holds the same parameters."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(N_GPUS), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "hmvit_amd.trainer", "--epochs", "3", "--frames", str(2 * N_GPUS), "--agents", "3",
           "--grid", "128", "96", "--small", "--backend", "nccl"]
    out = subprocess.run(cmd, env=_clean_env(), cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["world_size"] == N_GPUS and res["steps"] >= 3
    assert all(l == l and abs(l) < 1e6 for l in res["epoch_loss"]) and res["epoch_loss"][-1] < res["epoch_loss"][0]
    assert res["rank_param_spread"] <= 1e-6, res["rank_param_spread"]
