"""CPU, world_size 2, gloo: the data-parallel host logic (bucketed all-reduce-mean of the flat gradient
arena, initial broadcast) that runs over RCCL on the GPUs.  The kernels are not involved."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from real_esrgan_pytorch_amd.train import DataParallel
    dp = DataParallel(bucket_bytes=4 * 1000)          # force several buckets incl. a ragged last one
    assert dp.world == world
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_007, generator=g)
    ref = sum(torch.randn(10_007, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    dp.all_reduce_mean_(flat)
    ok1 = torch.allclose(flat, ref, atol=1e-6)
    w = torch.full((33,), float(rank))
    dp.broadcast_(w)
    ok2 = bool((w == 0).all())
    q.put((rank, ok1, ok2))
    dist.destroy_process_group()


def test_bucketed_allreduce_and_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res


def test_single_process_is_identity():
    from real_esrgan_pytorch_amd.train import DataParallel
    dp = DataParallel()
    t = torch.arange(10.0)
    dp.all_reduce_mean_(t)
    assert torch.equal(t, torch.arange(10.0))
