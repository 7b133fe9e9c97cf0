"""World-size-2 gloo test (CPU) of the multi-GPU bookkeeping bench.py uses: scene sharding covers
every scene exactly once, and the job time is the MAX over ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("hmvit_dist", os.path.join(root, "hm-vit_amd", "dist.py"))
    d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(d)
    mine = d.shard_scenes(7, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    t = d.max_over_ranks(1.0 + rank)              # rank 1 is the slow one
    thr = d.aggregate_throughput(len(mine), 1.0 + rank)   # 4 + 3 scenes
    dist.barrier()
    if rank == 0:
        out.put((gathered, t, thr))
    dist.destroy_process_group()


def test_two_rank_sharding_and_max_time():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, t, thr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(gathered[0] + gathered[1]) == list(range(7))
    assert not set(gathered[0]) & set(gathered[1])
    assert t == 2.0
    assert abs(thr - 7 / 2.0) < 1e-12


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher must start both ranks itself (fresh child processes) and print ONE
    JSON line from rank 0 with n_gpus = 2 -- the command the round-end driver may use.  CPU stand-in workload over gloo:
    this exercises bench.py's own spawn / rendezvous / barrier / max-over-ranks path, not the kernels."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--backend", "gloo", "--stub"], env=env, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["value"] > 0 and abs(r["value"] - 2 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-6


def test_bench_under_a_launcher_environment():
    """The torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, no self-spawn."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--backend", "gloo", "--stub"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1].decode()[-2000:]
    assert json.loads(outs[0][0].decode().strip().splitlines()[-1])["n_gpus"] == 2
    assert not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")]   # only rank 0 reports


def test_bench_train_mode_two_ranks_ddp():
    """`python bench.py --train --gpus 2`: the training half of north_star - one DistributedDataParallel step per rank with the
    gradient all-reduce as the only exchange (find_unused_parameters=True, train_camera.py:126-131).  CPU stand-in model (with
    an unused parameter) over gloo: exercises the spawn / DDP / no_sync / stand-alone all-reduce / reporting path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--train", "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--backend", "gloo", "--stub"], env=env, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["unit"] == "steps/s" and "dp2" in r["config"]["parallelism"]
    assert r["value"] > 0 and abs(r["value"] - 2 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-6
    assert r["gradient_bytes"] == 4 * (64 * 64 + 64)              # the unused Linear(8, 8) has no gradient
    assert r["allreduce_standalone_ms"] > 0 and r["ms_per_step_no_sync"] > 0 and r["allreduce_exposed_ms"] >= 0
