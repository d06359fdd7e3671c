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
    dp.all_reduce_mean_(flat)                           # one message (the default behind the backward pass)
    ok1 = torch.allclose(flat, ref, atol=1e-6)
    dp.bucketed = True                                  # ... and as bucket_bytes pieces ($RESR_DP_BUCKETED=1)
    flat2 = torch.randn(10_007, generator=torch.Generator().manual_seed(100 + rank))
    dp.all_reduce_mean_(flat2)
    ok1 = ok1 and torch.allclose(flat2, ref, atol=1e-6)
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


def _grads_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from real_esrgan_pytorch_amd.train import DataParallel, _arena_of
    dp = DataParallel(bucket_bytes=4 * 100)
    torch.manual_seed(5)
    # (a) gradients that are consecutive views of one arena (what the networks' backward passes hand out): reduced in place
    arena = torch.full((60,), float(rank + 1))
    ps = [torch.nn.Parameter(torch.zeros(4, 5)), torch.nn.Parameter(torch.zeros(10)), torch.nn.Parameter(torch.zeros(30))]
    off = 0
    for p in ps:
        p.grad = arena[off:off + p.numel()].view(p.shape)
        off += p.numel()
    assert _arena_of([p.grad for p in ps]) is not None
    dp.all_reduce_grads_(ps)
    ok_a = bool(torch.allclose(arena, torch.full((60,), 1.5)))
    # (b) scattered gradients (+ one parameter without a gradient): through a flat copy, written back
    qs = [torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(3, 3)), torch.nn.Parameter(torch.zeros(2))]
    qs[0].grad = torch.full((7,), float(rank))
    qs[1].grad = torch.full((3, 3), float(10 * rank))
    assert _arena_of([qs[0].grad, qs[1].grad]) is None
    dp.all_reduce_grads_(qs)
    ok_b = bool(torch.allclose(qs[0].grad, torch.full((7,), 0.5)) and torch.allclose(qs[1].grad, torch.full((3, 3), 5.0)) and qs[2].grad is None)
    # (c) attach_discriminator: parameters and buffers (spectral-norm u / v) follow rank 0

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.full((5,), float(rank)))
            self.register_buffer("weight_u", torch.full((3,), float(rank) + 7))

        def flat_parameters(self):
            return self.w.data
    m = Fake()
    dp.attach_discriminator(m)
    ok_c = bool((m.w == 0).all() and (m.weight_u == 7).all())
    q.put((rank, ok_a, ok_b, ok_c))
    dist.destroy_process_group()


def test_module_grad_exchange_and_buffer_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grads_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] and r[2] and r[3] for r in res), res


def _ranges_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import DataParallel
    g = R.Generator(3, 3, 4)                               # parameter containers only: no GPU involved
    ranges = g.grad_ranges()                               # the 25 ranges resr_generator_backward fires events for
    total = sum(p.numel() for p in g.parameters())
    dp = DataParallel()                                    # the real 12 MB buckets
    buckets = dp.merge_ranges(ranges)
    idx = torch.arange(total, dtype=torch.float32)
    flat = (idx % 1000) * 1e-3 + float(rank)
    dp.all_reduce_ranges_(flat, ranges, [None] * len(ranges))
    want = (idx % 1000) * 1e-3 + (world - 1) / 2.0
    q.put((rank, len(ranges), total, buckets, float((flat - want).abs().max())))
    dist.destroy_process_group()


def test_overlapped_exchange_buckets_world8_real_range_table():
    """World 8 (the node BASELINE config 4 names), gloo, host tensors: the bucket merging of `all_reduce_ranges_` over the REAL
    range table of the 23-block generator -- tail convs, RRDB 22 ... 0, conv1 -- reduces every element of the 16.7 M-element
    arena exactly once (a gap or an overlap between buckets would show as a wrong mean)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ranges_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == list(range(world))
    nr, total, buckets = res[0][1], res[0][2], res[0][3]
    assert nr == 25 and total == 16_697_987                     # 16,697,987 parameters
    # buckets: disjoint, descending, covering [0, total); all but the last at least 12 MB; each closes on the event of its last range
    assert buckets[0][1] == total and buckets[-1][0] == 0 and buckets[-1][2] == nr - 1
    for (lo, hi, last), nxt in zip(buckets, buckets[1:] + [None]):
        assert hi > lo and (nxt is None or nxt[1] == lo)
        assert nxt is None or (hi - lo) * 4 >= 12 << 20
    assert 4 <= len(buckets) <= 6, buckets                        # 66.8 MB in >= 12 MB buckets
    assert all(r[4] < 1e-5 for r in res), [r[4] for r in res]


def _forced_world1_worker(port, q):
    """`DataParallel(force=True)` with a world of ONE rank: every collective branch runs (this is how the single-GPU tests and
    `RESR_BENCH_FORCE_NCCL=1 bench.py` execute the RCCL path); with gloo on the CPU the host logic of the same branches."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=0, world_size=1)
        from real_esrgan_pytorch_amd.train import DataParallel
        assert not DataParallel().active                      # a world of one is inactive unless forced
        os.environ["RESR_DP_FORCE"] = "1"
        assert DataParallel().active                          # ... by the environment
        os.environ.pop("RESR_DP_FORCE")
        dp = DataParallel(bucket_bytes=4 * 1000, force=True)
        assert dp.active and dp.world == 1 and not dp._avg     # gloo sums and scales; nccl averages inside the collective
        flat = torch.randn(10_007, generator=torch.Generator().manual_seed(1))
        ref = flat.clone()
        dp.all_reduce_mean_(flat)
        ok = torch.equal(flat, ref)
        ranges = [(8000, 10_007), (5000, 8000), (1200, 5000), (0, 1200)]
        dp.all_reduce_ranges_(flat, ranges, [None] * 4)
        ok = ok and torch.equal(flat, ref)
        p = torch.nn.Parameter(torch.zeros(7))
        p.grad = torch.arange(7.0)
        dp.all_reduce_grads_([p])
        ok = ok and torch.equal(p.grad, torch.arange(7.0))

        class E:
            _flat_shadow = torch.arange(5.0)
            shadow = {}
        dp.attach_ema(E)
        dist.destroy_process_group()
        q.put(bool(ok))
    except Exception as e:  # pragma: no cover
        q.put(repr(e))


def test_forced_collectives_with_a_world_of_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_world1_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert res is True, res


def test_graphed_step_refuses_what_it_cannot_capture():
    """train.GraphedStep: optimisers must be capturable (device-side step counters) and the step must not hold an active
    data-parallel exchange -- checked at construction, on the host."""
    import pytest
    from real_esrgan_pytorch_amd.train import GraphedStep

    class Step:
        def __init__(self, opt, dp=None):
            self.optimizer, self.dp = opt, dp
    w = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(ValueError):
        GraphedStep(Step(torch.optim.Adam([w], 1e-3)))

    class DP:
        active = True
    with pytest.raises(ValueError):
        GraphedStep(Step(torch.optim.Adam([w], 1e-3, capturable=True), DP()))
    GraphedStep(Step(torch.optim.Adam([w], 1e-3, capturable=True)))


def test_loss_helpers_call_a_non_stock_criterion():
    """losses.l1_loss / bce_with_logits_const fuse ONLY the stock criteria in their default configuration; anything else (and any
    tensor the fused launch does not take -- here: CPU tensors) is handed to the criterion as the reference's loop would."""
    from real_esrgan_pytorch_amd import losses
    a, b = torch.rand(2, 3, 4, 4), torch.rand(2, 3, 4, 4)
    assert torch.allclose(losses.l1_loss(torch.nn.SmoothL1Loss(), a, b, 0.5), 0.5 * torch.nn.functional.smooth_l1_loss(a, b))
    assert torch.allclose(losses.l1_loss(torch.nn.L1Loss(), a, b), torch.nn.functional.l1_loss(a, b))
    x = torch.randn(2, 1, 4, 4)
    want = 0.1 * torch.nn.functional.binary_cross_entropy_with_logits(x, torch.ones_like(x))
    assert torch.allclose(losses.bce_with_logits_const(torch.nn.BCEWithLogitsLoss(), x, 1.0, 0.1), want)
    summed = losses.bce_with_logits_const(torch.nn.BCEWithLogitsLoss(reduction="sum"), x, 0.0)
    assert torch.allclose(summed, torch.nn.functional.binary_cross_entropy_with_logits(x, torch.zeros_like(x), reduction="sum"))


def test_bench_plain_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts two ranks through torch.distributed.run
    (children, never an exec) and hands back their exit code.  Without a GPU each rank refuses loudly (the hot path has no CPU
    fallback), so here the relayed code is non-zero and the refusal is what stderr shows -- the launch path itself is what runs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert "starting 2 ranks" in r.stderr and "--nproc-per-node=2" in r.stderr
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs an MI355X" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a launcher whose world disagrees with --gpus is still an error, not a second launch
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, cwd=root, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "disagree" in r2.stderr and "starting" not in r2.stderr
