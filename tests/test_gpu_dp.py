"""-m gpu: data-parallel step with the real kernels.  Two processes share the one GPU of the test box (gloo
backend, since RCCL refuses two ranks on one device); each runs the generator on its half of the batch, the
flat-gradient all-reduce(mean) must reproduce the single-process gradient of the whole batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(0)
    return R.Generator(3, 3, 4, precision="strict", n_blocks=1).cuda()


def _data():
    g = torch.Generator().manual_seed(9)
    return torch.rand(4, 3, 16, 16, generator=g), torch.rand(4, 3, 64, 64, generator=g)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from real_esrgan_pytorch_amd.train import DataParallel
    torch.cuda.set_device(0)
    model = _model()
    if rank == 1:                                   # replicas must be made identical by the initial broadcast
        with torch.no_grad():
            model.flat_parameters().add_(1.0)
    DataParallel(bucket_bytes=64 << 10).attach(model)
    lr, hr = _data()
    half = slice(rank * 2, rank * 2 + 2)
    loss = (model(lr[half].cuda()) - hr[half].cuda()).abs().mean()
    loss.backward()
    torch.cuda.synchronize()
    q.put((rank, model.flat_grad().cpu().numpy(), model.flat_parameters().detach().cpu().numpy()))   # by value
    dist.destroy_process_group()


def test_two_rank_dp_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, g, w = q.get(timeout=600)
        res[r] = (torch.from_numpy(g), torch.from_numpy(w))
    for p in procs:
        p.join(timeout=120)
    model = _model()
    lr, hr = _data()
    (model(lr.cuda()) - hr.cuda()).abs().mean().backward()
    ref = model.flat_grad().cpu()
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], model.flat_parameters().detach().cpu())   # broadcast
    assert torch.equal(res[0][0], res[1][0])                                                            # same reduced gradient
    rel = ((res[0][0] - ref).norm() / ref.norm()).item()
    assert rel < 1e-5, rel                    # fp32: mean of two half-batch gradients == whole-batch gradient


def test_grad_ready_events_fire_after_their_range_is_final():
    """resr_generator_backward's grad_ready_events: a side stream that waits for event i and copies arena range i must see the
    FINAL gradients of that range (that is what lets the all-reduce of a bucket run under the rest of the backward pass), the
    ranges tile the arena in backward order, and the gradients equal those of a backward without events."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=3).cuda()
    lr, hr = _data()
    (g(lr.cuda()) - hr.cuda()).abs().mean().mul(1024.0).backward()
    torch.cuda.synchronize()
    ref = g.flat_grad().clone()
    ranges = g.grad_ranges()
    assert len(ranges) == 3 + 2 and ranges[0][1] == ref.numel() and ranges[-1][0] == 0
    assert all(ranges[i][0] == ranges[i + 1][1] for i in range(len(ranges) - 1))          # adjacent, descending
    side = torch.cuda.Stream()
    snap = {}

    def hook(flat, rngs, events):
        assert len(events) == len(rngs) == 5
        for i, ((lo, hi), ev) in enumerate(zip(rngs, events)):
            side.wait_event(ev)
            with torch.cuda.stream(side):
                snap[i] = flat[lo:hi].clone()
        torch.cuda.current_stream().wait_stream(side)
    g.grad_ready_hook = hook
    g.zero_grad(set_to_none=True)
    g.flat_grad().zero_()
    (g(lr.cuda()) - hr.cuda()).abs().mean().mul(1024.0).backward()
    torch.cuda.synchronize()
    assert torch.equal(g.flat_grad(), ref)
    for i, (lo, hi) in enumerate(ranges):
        assert torch.equal(snap[i], ref[lo:hi]), i


def _gan_setup(precision="strict"):
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision=precision, n_blocks=1).cuda()
    d = R.Discriminator(precision=precision).cuda()
    return R, g, d


def _gan_step(R, g, d, lr, hr, dp):
    from real_esrgan_pytorch_amd.train import RealESRGANStep
    g.train(); d.train()
    ema = R.EMA(g, 0.999)
    ema.register()
    # learning rate 0: the step runs end to end (both optimisers, EMA) and leaves the reduced gradients in place
    step = RealESRGANStep(g, d, ema, torch.optim.Adam(g.parameters(), 0.0, (0.9, 0.99)),
                          torch.optim.Adam(d.parameters(), 0.0, (0.9, 0.99)), scaler=None, dp=dp)
    out = step(hr.cuda(), lr.cuda())
    torch.cuda.synchronize()
    gd = torch.cat([p.grad.reshape(-1) for p in d.parameters()]).cpu()
    return g.flat_grad().cpu(), gd, {k: float(v) for k, v in out.items()}


def _gan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from real_esrgan_pytorch_amd.train import DataParallel
    torch.cuda.set_device(0)
    R, g, d = _gan_setup()
    if rank == 1:                                   # replicas (weights AND spectral-norm u / v) must be made identical by attach
        with torch.no_grad():
            g.flat_parameters().add_(0.5)
            d.flat_parameters().mul_(1.5)
            for b in d.buffers():
                b.add_(0.1)
    dp = DataParallel(bucket_bytes=256 << 10)
    dp.attach(g)
    dp.attach_discriminator(d)
    lr, hr = _data()
    half = slice(rank * 2, rank * 2 + 2)
    gg, gd, losses = _gan_step(R, g, d, lr[half], hr[half], dp)
    u = torch.cat([b.reshape(-1) for b in d.buffers()]).cpu()
    q.put((rank, gg.numpy(), gd.numpy(), u.numpy(), losses))
    dist.destroy_process_group()


def test_two_rank_gan_step_matches_single_process():
    """Config 4's data-parallel leg (reference step: train_realesrgan.py:459-521): generator AND discriminator gradients of two
    ranks on half batches, after the exchanges, equal the single-process gradients of the whole batch; the discriminator's two
    backwards are reduced by ONE exchange after the second; u / v stay identical across ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_gan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r, gg, gd, u, losses = q.get(timeout=900)
        res[r] = (torch.from_numpy(gg), torch.from_numpy(gd), torch.from_numpy(u), losses)
    for p in procs:
        p.join(timeout=120)
    R, g, d = _gan_setup()
    lr, hr = _data()
    gg, gd, losses = _gan_step(R, g, d, lr, hr, None)
    u = torch.cat([b.reshape(-1) for b in d.buffers()]).cpu()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])          # both ranks hold the same reduced gradients
    assert torch.allclose(res[0][2], res[1][2], atol=1e-6) and torch.allclose(res[0][2], u, atol=1e-5)   # u / v: broadcast, then 3 identical power iterations
    rel_g = ((res[0][0] - gg).norm() / gg.norm()).item()
    rel_d = ((res[0][1] - gd).norm() / gd.norm()).item()
    assert rel_g < 1e-5 and rel_d < 1e-5, (rel_g, rel_d)
    for k in ("pixel_loss", "adversarial_loss", "d_loss_hr", "d_loss_sr"):                  # batch means: mean of the two halves
        assert abs(0.5 * (res[0][3][k] + res[1][3][k]) - losses[k]) < 1e-5, k


def test_bench_two_ranks_prints_one_line():
    """The driver's multi-GPU launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) with two
    ranks sharing the test box's one GPU (gloo): warm-up, timed steps, the in-situ roofline step (a collective step: every
    rank must run it) and the teardown complete, and rank 0 prints exactly one JSON line with the aggregate rate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RESR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--lr-size", "64", "--no-parity-mode", "--no-sustained"]   # (the exact16 sub-runs of the line are single-rank matters: 40 s of this test)
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["scaling"] == "weak" and "roofline" in out and "cpu_baseline" not in out
    assert "error" not in out["roofline"], out["roofline"]
    # self-diagnosing line: which devices the ranks drove, what a step costs the host, both exchange policies timed
    d = out["dist"]
    assert d["nranks"] == 2 and d["devices"] == 1 and len(d["device_list"]) == 2 and "SHARE" in d["warning"]
    assert set(d["policies"]) >= {"sequential", "overlap_31cu", "chosen"} and d["policies"]["chosen"] in ("sequential", "overlap_31cu")
    best = min(d["policies"]["sequential"]["ms_per_step"], d["policies"]["overlap_31cu"]["ms_per_step"])
    assert abs(out["ms_per_step"] - best) < 0.02, (out["ms_per_step"], d["policies"])
    assert out["host"]["enqueue_ms_per_step"] > 0 and out["host"]["cpus_in_affinity"] >= 1


def test_bench_gan_two_ranks_plain_launch_prints_one_line():
    """PLAIN `python bench.py --gpus 2 --gan` -- no launcher, no WORLD_SIZE in the environment (BASELINE config 4): bench.py starts
    its two ranks itself as children of a process that never touches the GPU (`self_launch`), generator + discriminator exchanges
    run, and exactly one JSON line with `dist.nranks == 2` comes back through the parent's stdout with the children's exit code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RESR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--lr-size", "32", "--gan", "--no-parity-mode", "--no-sustained"]   # (LR 16: a resize1 factor of 0.15 leaves 9 pixels for the 21-tap blur, which reflect padding refuses -- as F.pad does)
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "starting 2 ranks" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["dist"]["nranks"] == 2 and out["dist"]["world"] == 2
    assert out["value"] > 0 and "GAN" in out["metric"] and set(out["losses"]) >= {"pixel_loss", "adversarial_loss", "d_loss_hr", "d_loss_sr", "content_loss"}
    assert "roofline" in out and "error" not in out["roofline"], out.get("roofline")


def test_bench_gan_eight_ranks_on_one_gpu():
    """The SCALE run's widest launch line (`--nproc-per-node 8 ... bench.py --gpus 8 --gan`, BASELINE config 4: global batch
    8 x per-GPU batch) at a tiny size with the eight ranks sharing the test box's one GPU (gloo; LOCAL_RANK % device_count):
    ports, seeds, both exchange policies, the roofline step and the teardown complete; ONE JSON line whose `dist` record says
    that eight ranks ran on ONE distinct device."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RESR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--batch", "1", "--lr-size", "32", "--gan", "--no-sustained"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp8"
    d = out["dist"]
    assert d["world"] == d["nranks"] == 8 and d["devices"] == 1 and [x["rank"] for x in d["device_list"]] == list(range(8))
    assert d["policies"]["chosen"] in ("sequential", "overlap_31cu") and "overlap_31cu" in d["policies"]
    assert out["host"]["enqueue_ms_per_step"] >= out["host"]["enqueue_ms_min_rank"] > 0
    assert out["chain_errors"] == 0


def test_train_script_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 ... train_realesrnet` (reference train_realesrnet.py:105-129 loop,
    one process per GPU; here two ranks on the box's one GPU, gloo): the DistributedSampler shards the epoch (the ranks see
    disjoint images, together all of them), the ranks draw different degradations, only rank 0 writes checkpoints, and after
    the epoch's three optimiser steps both replicas hold bit-identical weights."""
    import json
    import subprocess
    import sys
    import numpy as np
    from PIL import Image
    from tests.test_gpu_train_harness import _smooth_png
    from real_esrgan_pytorch_amd import imgproc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub, n in (("train", 12), ("valid", 2), ("test_hr", 2)):
        os.makedirs(tmp_path / sub)
        for i in range(n):
            _smooth_png(str(tmp_path / sub / f"{i}.png"), 224, 7 * len(sub) + i)
    os.makedirs(tmp_path / "test_lr")
    for i in range(2):
        hr = imgproc.read_image_rgb(str(tmp_path / "test_hr" / f"{i}.png"))
        Image.fromarray((np.clip(imgproc.image_resize(hr, 0.25), 0, 1) * 255).round().astype(np.uint8)).save(str(tmp_path / "test_lr" / f"{i}.png"))
    driver = tmp_path / "driver.py"
    driver.write_text(f"""
import json, os, sys, hashlib
import torch
sys.path.insert(0, {root!r})
from real_esrgan_pytorch_amd import config
from real_esrgan_pytorch_amd import train_realesrnet as T
tmp = {str(tmp_path)!r}
for k, v in dict(train_image_dir=tmp + "/train", valid_image_dir=tmp + "/valid", test_lr_image_dir=tmp + "/test_lr",
                 test_hr_image_dir=tmp + "/test_hr", image_size=208, batch_size=2, num_workers=0, epochs=1, print_frequency=1,
                 resume="", lr_scheduler_step_size=1, exp_name="dp_test", precision="fast",
                 niqe_model_path={os.path.join(root, "tests", "golden", "niqe_model.mat")!r}).items():
    setattr(config, k, v)
os.chdir(tmp)
rank = int(os.environ["RANK"])
log = dict(rank=rank, hr=[], lr=[])
Base = T.RealESRNetStep
class Step(Base):
    def __call__(self, hr, lr=None):
        log["hr"].append(hashlib.sha1(hr.cpu().numpy().tobytes()).hexdigest())
        log["lr"].append(float(lr.float().mean()))
        return super().__call__(hr, lr)
T.RealESRNetStep = Step
validate = T.validate
def validate_and_dump(model, ema_model, *a, **k):
    if "weights" not in log:
        torch.cuda.synchronize()
        log["weights"] = hashlib.sha1(model.flat_parameters().detach().cpu().numpy().tobytes()).hexdigest()
        log["ema"] = hashlib.sha1(torch.cat([ema_model.shadow[n].reshape(-1) for n in sorted(ema_model.shadow)]).cpu().numpy().tobytes()).hexdigest()
    out = validate(model, ema_model, *a, **k)
    log.setdefault("niqe", []).append(float(out))
    return out
T.validate = validate_and_dump
T.main()
json.dump(log, open(tmp + f"/log_rank{{rank}}.json", "w"))
""")
    env = dict(os.environ, RESR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), str(driver)]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    logs = [json.load(open(tmp_path / f"log_rank{k}.json")) for k in (0, 1)]
    assert all(len(l["hr"]) == 3 for l in logs)                                  # 12 images / 2 ranks / batch 2
    assert not set(logs[0]["hr"]) & set(logs[1]["hr"])                          # disjoint shards
    assert logs[0]["lr"] != logs[1]["lr"]                                        # rank-dependent degradation draws
    assert logs[0]["weights"] == logs[1]["weights"]                              # replicas stay bit-identical
    assert logs[0]["ema"] == logs[1]["ema"]                                      # ... and so do their EMA shadows (rank 0's init, then identical updates)
    assert logs[0]["niqe"][1] == logs[1]["niqe"][1]                              # the "Test" evaluation (same images on every rank) scores the same model
    assert (tmp_path / "samples" / "dp_test" / "g_epoch_1.pth.tar").exists()    # written once, by rank 0
    tags = [json.loads(l) for l in open(tmp_path / "samples" / "logs" / "dp_test" / "scalars.jsonl")]
    assert sum(t["tag"] == "Train/Loss" for t in tags) == 3                     # one writer


# ---- the RCCL path itself, on the one GPU of the test box ------------------------------------------------------------------
# RCCL refuses two ranks on one device, so the two-rank tests above use gloo.  A WORLD-1 `nccl` group, however, launches genuine
# RCCL kernels: with DataParallel(force=True) every collective branch runs (warm-up, broadcast, ReduceOp.AVG on arena slices,
# the communication stream behind the backward pass's range events) -- next to the chained dense-block launches that want
# every CU.  World 1 makes the expected result exact: the averaged gradient IS the local gradient, bit for bit.
def _nccl_world1_worker(port, q, overlap):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import real_esrgan_pytorch_amd as R
        from real_esrgan_pytorch_amd.train import DataParallel, RealESRGANStep

        def make():
            torch.manual_seed(0)
            m = R.Generator(3, 3, 4, precision="fast", n_blocks=3).cuda()
            with torch.no_grad():
                m.conv4.bias.add_(0.5)
            return m
        gen = torch.Generator().manual_seed(4)
        lr = torch.rand(8, 3, 64, 64, generator=gen).cuda()      # batch % 8 == 0: the dense blocks run as chained launches
        hr = torch.rand(8, 3, 256, 256, generator=gen).cuda()
        ref_m = make()
        (ref_m(lr) - hr).abs().mean().mul(1024.0).backward()
        torch.cuda.synchronize()
        ref = ref_m.flat_grad().clone()
        m = make()
        dp = DataParallel(bucket_bytes=256 << 10, force=True)
        assert dp.active and dp.backend == "nccl" and dp._avg
        dp.attach(m, overlap=overlap)
        assert dp.overlap == overlap
        calls = []
        if overlap:
            inner = m.grad_ready_hook

            def spy(flat, ranges, events):
                calls.append(len(dp.merge_ranges(ranges)))
                return inner(flat, ranges, events)
            m.grad_ready_hook = spy
        outs = []
        for _ in range(3):                                       # RCCL kernels of step i's exchange next to step i+1's chains
            m.zero_grad(set_to_none=True)
            (m(lr) - hr).abs().mean().mul(1024.0).backward()
            outs.append(m.flat_grad().clone())
        torch.cuda.synchronize()
        res = {"bit_equal": all(torch.equal(o, ref) for o in outs), "chain_errors": int(R._lib.lib().resr_debug_chain_errors()),
               "buckets": calls, "comm_stream": getattr(dp, "_comm", None) is not None,
               "weights_equal": torch.equal(m.flat_parameters(), ref_m.flat_parameters())}
        # the discriminator leg: one GAN step with its single exchange after the second backward (train_realesrgan.py:503-516)
        torch.manual_seed(1)
        d = R.Discriminator(precision="fast").cuda().train()
        d_ref_grads = None
        for use_dp in (False, True):
            torch.manual_seed(2)
            g2 = make()
            d2 = R.Discriminator(precision="fast").cuda().train()
            d2.load_state_dict(d.state_dict())
            dp2 = None
            if use_dp:
                dp2 = DataParallel(force=True)
                dp2.attach(g2, overlap=overlap)
                dp2.attach_discriminator(d2)
            step = RealESRGANStep(g2, d2, None, torch.optim.SGD(g2.parameters(), 0.0), torch.optim.SGD(d2.parameters(), 0.0),
                                  scaler=torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=10 ** 9), dp=dp2)
            step(hr, lr)
            torch.cuda.synchronize()
            grads = torch.cat([p.grad.reshape(-1) for p in d2.parameters()] + [g2.flat_grad()])
            if not use_dp:
                d_ref_grads = grads
            else:
                res["gan_bit_equal"] = bool(torch.equal(grads, d_ref_grads))
        res["chain_errors_after_gan"] = int(R._lib.lib().resr_debug_chain_errors())
        dist.destroy_process_group()
        q.put(res)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put({"error": repr(e), "trace": traceback.format_exc()})


@pytest.mark.parametrize("overlap", [True, False])
def test_rccl_world1_collectives_next_to_chained_launches(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_port(), q, overlap))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert "error" not in res, res
    assert res["bit_equal"] and res["weights_equal"], res          # ReduceOp.AVG over one rank: the gradient itself
    assert res["gan_bit_equal"], res
    assert res["chain_errors"] == 0 and res["chain_errors_after_gan"] == 0, res
    if overlap:
        assert res["comm_stream"] and res["buckets"] and all(b >= 2 for b in res["buckets"]), res   # several buckets behind range events


@pytest.mark.parametrize("overlap_env", ["1", "0"])
def test_bench_force_nccl_one_gpu(overlap_env):
    """RESR_BENCH_FORCE_NCCL=1 python bench.py --gpus 1: the bench's own steps with a world-1 RCCL group (dist.backend "nccl").
    overlap_env "0": the default environment of a SCALE run -- BOTH exchange policies are timed in one process with real RCCL
    kernels (sequential, then overlapped with 31 CUs per XCD left to the chained launches), the better one is `value`."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RESR_BENCH_FORCE_NCCL="1", RESR_DP_OVERLAP=overlap_env, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--lr-size", "64", "--no-other-configs", "--no-cpu-baseline", "--no-parity-mode"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["dist"]["backend"] == "nccl" and out["dist"]["forced_collectives"] and out["dist"]["overlap_with_backward"] == (overlap_env == "1")
    assert out["chain_errors"] == 0 and out["value"] > 0
    pol = out["dist"]["policies"]
    if overlap_env == "1":
        assert pol["chosen"] == "overlap_env"
    else:
        assert set(pol) >= {"sequential", "overlap_31cu"} and pol["chosen"] in ("sequential", "overlap_31cu")
        assert abs(out["ms_per_step"] - min(pol["sequential"]["ms_per_step"], pol["overlap_31cu"]["ms_per_step"])) < 0.02
    assert out["dist"]["devices"] == 1 and out["dist"]["nranks"] == 1 and out["host"]["enqueue_ms_per_step"] > 0
