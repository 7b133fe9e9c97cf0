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
    thr = d.aggregate_throughput(10, 1.0 + rank)
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
    assert abs(thr - 2 * 10 / 2.0) < 1e-12
