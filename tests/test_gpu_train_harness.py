"""Training entry point of SURVEY §8f rank 4 on a tiny synthetic dataset: two epochs of
`real_esrgan_pytorch_amd.train_realesrnet.main()` with dataset-driven degradation, NIQE validation under the EMA
weights, the reference's checkpoint dictionary / file names, and a resume."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _smooth_png(path, size, seed):
    from PIL import Image
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x = F.interpolate(torch.rand(1, 3, size // 8, size // 8, generator=g), size=(size, size), mode="bicubic").clamp(0, 1)
    x = (0.85 * x + 0.15 * torch.rand(1, 3, size, size, generator=g)).clamp(0, 1)
    Image.fromarray((x[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(path)


def _tiny_dataset(tmp_path, monkeypatch, **extra):
    from real_esrgan_pytorch_amd import config, imgproc
    for sub, n in (("train", 4), ("valid", 2), ("test_hr", 2)):
        os.makedirs(tmp_path / sub)
        for i in range(n):
            _smooth_png(str(tmp_path / sub / f"{i}.png"), 224, hash(sub) % 100 + i)
    os.makedirs(tmp_path / "test_lr")
    from PIL import Image
    for i in range(2):
        hr = imgproc.read_image_rgb(str(tmp_path / "test_hr" / f"{i}.png"))
        lr = np.clip(imgproc.image_resize(hr, 0.25), 0, 1)
        Image.fromarray((lr * 255).round().astype(np.uint8)).save(str(tmp_path / "test_lr" / f"{i}.png"))
    here = os.path.dirname(os.path.abspath(__file__))
    for k, v in dict(train_image_dir=str(tmp_path / "train"), valid_image_dir=str(tmp_path / "valid"),
                     test_lr_image_dir=str(tmp_path / "test_lr"), test_hr_image_dir=str(tmp_path / "test_hr"),
                     image_size=208, batch_size=2, num_workers=0, epochs=1, print_frequency=1, resume="",
                     lr_scheduler_step_size=1, exp_name="harness_test",
                     niqe_model_path=os.path.join(here, "golden", "niqe_model.mat"),
                     device=torch.device("cuda", 0), **extra).items():
        monkeypatch.setattr(config, k, v, raising=False)
    monkeypatch.chdir(tmp_path)


def test_train_validate_checkpoint_resume(tmp_path, monkeypatch):
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd import config
    from real_esrgan_pytorch_amd import train_realesrnet as T
    _tiny_dataset(tmp_path, monkeypatch)
    T.main()
    ck1 = tmp_path / "samples" / "harness_test" / "g_epoch_1.pth.tar"
    assert ck1.exists() and (tmp_path / "results" / "harness_test" / "g_last.pth.tar").exists()
    assert (tmp_path / "results" / "harness_test" / "g_best.pth.tar").exists()          # first NIQE < 100
    ck = torch.load(ck1, weights_only=False)
    assert set(ck) == {"epoch", "best_niqe", "state_dict", "ema_state_dict", "optimizer", "scheduler"}
    assert ck["epoch"] == 1 and len(ck["state_dict"]) == 702 and np.isfinite(ck["best_niqe"])
    assert all(k.startswith("model.") for k in ck["ema_state_dict"])                     # reference key prefix (inference.py:33)
    tags = [json.loads(l)["tag"] for l in open(tmp_path / "samples" / "logs" / "harness_test" / "scalars.jsonl")]
    assert "Train/Loss" in tags and "Valid/NIQE" in tags and "Test/NIQE" in tags
    # resume for one more epoch
    monkeypatch.setattr(config, "resume", str(ck1))
    monkeypatch.setattr(config, "epochs", 2)
    T.main()
    ck2 = torch.load(tmp_path / "samples" / "harness_test" / "g_epoch_2.pth.tar", weights_only=False)
    assert ck2["epoch"] == 2 and ck2["optimizer"]["state"][0]["step"] == 4               # 2 steps per epoch, state carried over
    # the reference's inference loader reads it: strip "model." (inference.py:33)
    g = R.Generator(3, 3, 4).cuda()
    g.load_state_dict({k[len("model."):]: v for k, v in ck2["ema_state_dict"].items()})


def test_degradation_runs_one_batch_ahead_on_a_side_stream(tmp_path, monkeypatch):
    """What the train scripts run is what bench.py times: the degradation kernels of batch i+1 are enqueued on a side stream
    BEFORE step i is issued and complete while step i is still running (reference slot: CUDAPrefetcher, dataset.py:271-312)."""
    from real_esrgan_pytorch_amd import train_realesrnet as T
    _tiny_dataset(tmp_path, monkeypatch)
    made, seen = [], []

    class Pre(T.DegradationPrefetcher):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)

    class Step(T.RealESRNetStep):
        def __call__(self, hr, lr=None):
            assert lr is not None                       # the loop hands the step an already degraded batch
            ahead = made[-1].done_events if made[-1]._pending is not None else None   # batch i+1: enqueued before this step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = super().__call__(hr, lr)
            e1.record()
            seen.append((e0, e1, ahead))
            return out

    monkeypatch.setattr(T, "DegradationPrefetcher", Pre)
    monkeypatch.setattr(T, "RealESRNetStep", Step)
    T.main()
    torch.cuda.synchronize()
    assert len(made) == 1 and made[0].stream != torch.cuda.default_stream() and made[0].stream != torch.cuda.current_stream()
    assert len(seen) == 2 and seen[0][2] is not None and seen[1][2] is None      # 4 images, batch 2: step 0 has a batch ahead of it
    e0, e1, (d0, d1) = seen[0]
    step_ms, deg_end_ms = e0.elapsed_time(e1), e0.elapsed_time(d1)
    assert d0.elapsed_time(d1) > 0 and deg_end_ms < step_ms, (step_ms, deg_end_ms)   # batch 1's degradation finished under step 0


def test_gan_train_checkpoints_resume_and_directory_test(tmp_path, monkeypatch):
    """`train_realesrgan.main()` from a RealESRNet checkpoint, its two checkpoint families, resume_d / resume_g, and
    `test.main()` over the LR folder with the resulting EMA weights."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd import config
    from real_esrgan_pytorch_amd import test as E
    from real_esrgan_pytorch_amd import train_realesrgan as T
    _tiny_dataset(tmp_path, monkeypatch, resume_d="", resume_g="", pixel_weight=1.0, adversarial_weight=0.1,
                  content_weight=[0.1, 0.1, 1.0, 1.0, 1.0], lr_scheduler_milestones=[1], lr_scheduler_gamma=0.5,
                  model_lr=1e-4, model_betas=(0.9, 0.99), ema_model_weight_decay=0.999,
                  feature_model_extractor_nodes=["features.2", "features.7", "features.16", "features.25", "features.34"],
                  feature_model_normalize_mean=[0.485, 0.456, 0.406], feature_model_normalize_std=[0.229, 0.224, 0.225])
    torch.manual_seed(3)
    net = R.Generator(3, 3, 4)
    torch.save({"state_dict": net.state_dict()}, tmp_path / "esrnet.pth.tar")            # what train_realesrnet leaves (g_last)
    monkeypatch.setattr(config, "resume", str(tmp_path / "esrnet.pth.tar"))
    T.main()
    samples, results = tmp_path / "samples" / "harness_test", tmp_path / "results" / "harness_test"
    for name in ("d_best", "g_best", "d_last", "g_last"):
        assert (results / f"{name}.pth.tar").exists()
    d1 = torch.load(samples / "d_epoch_1.pth.tar", weights_only=False)
    g1 = torch.load(samples / "g_epoch_1.pth.tar", weights_only=False)
    assert set(d1) == {"epoch", "best_niqe", "state_dict", "optimizer", "scheduler"}
    assert set(g1) == {"epoch", "best_niqe", "state_dict", "ema_state_dict", "optimizer", "scheduler"}
    assert any(k.endswith("weight_orig") for k in d1["state_dict"])                      # spectral-norm parametrisation kept
    w0, w1 = net.state_dict()["conv1.weight"], g1["state_dict"]["conv1.weight"].cpu()
    assert not torch.equal(w0, w1) and (w0 - w1).abs().max() < 1e-2                      # started from the RealESRNet weights, then moved
    tags = {json.loads(l)["tag"] for l in open(tmp_path / "samples" / "logs" / "harness_test" / "scalars.jsonl")}
    assert {"Train/D_Loss", "Train/G_Loss", "Train/Pixel_Loss", "Train/Content_Loss", "Train/Adversarial_Loss",
            "Train/D(HR)_Probability", "Train/D(SR)_Probability", "Valid/NIQE", "Test/NIQE"} <= tags
    # resume both networks for one more epoch
    for k, v in dict(resume="", resume_d=str(samples / "d_epoch_1.pth.tar"), resume_g=str(samples / "g_epoch_1.pth.tar"),
                     epochs=2).items():
        monkeypatch.setattr(config, k, v)
    T.main()
    d2 = torch.load(samples / "d_epoch_2.pth.tar", weights_only=False)
    g2 = torch.load(samples / "g_epoch_2.pth.tar", weights_only=False)
    assert d2["epoch"] == g2["epoch"] == 2
    assert d2["optimizer"]["state"][0]["step"] == 4 and g2["optimizer"]["state"][0]["step"] == 4
    assert g2["optimizer"]["param_groups"][0]["lr"] == pytest.approx(5e-5)        # 1e-4, milestone 1 passed once
    # directory evaluation with the EMA weights (reference test.py)
    for k, v in dict(lr_dir=str(tmp_path / "test_lr"), sr_dir=str(tmp_path / "sr"), hr_dir=str(tmp_path / "test_hr"),
                     model_path=str(results / "g_last.pth.tar")).items():
        monkeypatch.setattr(config, k, v, raising=False)
    score = E.main()
    assert 0 < score <= 100 and sorted(os.listdir(tmp_path / "sr")) == ["0.png", "1.png"]
    from PIL import Image
    assert Image.open(tmp_path / "sr" / "0.png").size == (224, 224)                     # 56x56 LR, x4



@pytest.mark.parametrize("gan", [False, True])
def test_graphed_step_equals_eager_step(gan):
    """train.GraphedStep: everything after the degradation replayed from one hipGraph (chained dense-block launches, fused losses,
    GradScaler bookkeeping, capturable fused Adam, EMA) -- losses of every step and the final weights bit-equal to the eager step."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import GraphedStep, RealESRGANStep, RealESRNetStep

    def run(graph):
        torch.manual_seed(0)
        g = R.Generator(3, 3, 4, n_blocks=2).cuda().train()
        with torch.no_grad():
            g.conv4.bias.add_(0.5)
        ema = R.EMA(g, 0.999)
        ema.register()
        go = torch.optim.Adam([g.flat_parameter()], 1e-4, (0.9, 0.99), fused=True, capturable=True)
        gen = torch.Generator(device="cuda").manual_seed(1)
        hr = torch.rand(8, 3, 128, 128, device="cuda", generator=gen)      # batch 8: chained launches inside the graph
        lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode="area")
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        d = None
        if gan:
            d = R.Discriminator().cuda().train()
            do = torch.optim.Adam([d.flat_parameter()], 1e-4, (0.9, 0.99), fused=True, capturable=True)
            step = RealESRGANStep(g, d, ema, go, do, scaler, None)
        else:
            step = RealESRNetStep(g, ema, go, scaler, None)
        if graph:
            step = GraphedStep(step, warmup=2)
        outs = []
        for _ in range(6):
            o = step(hr, lr)
            outs.append(torch.stack([o[k] for k in sorted(o)]).clone() if isinstance(o, dict) else o.clone())
        torch.cuda.synchronize()
        return outs, [g.flat_parameters().detach().clone()] + ([d.flat_parameters().detach().clone()] if gan else []), ema._flat_shadow.clone()
    oe, we, se = run(False)
    og, wg, sg = run(True)
    for a, b in zip(oe, og):
        assert torch.equal(a, b)
    for a, b in zip(we, wg):
        assert torch.equal(a, b)
    assert torch.equal(se, sg)
    assert int(R._lib.lib().resr_debug_chain_errors()) == 0
    with pytest.raises(ValueError):          # the step counter must live on the device
        g = R.Generator(3, 3, 4, n_blocks=1).cuda()
        GraphedStep(RealESRNetStep(g, None, torch.optim.Adam([g.flat_parameter()], 1e-4, fused=True), None, None))


@pytest.mark.parametrize("flat", [True, False])
def test_gradscaler_overflow_skips_the_realesrnet_step(flat):
    """A forced overflow (reference train_realesrnet.py:388-391: scaler.scale(loss).backward(); scaler.step(); scaler.update()): the
    loss scale starts far beyond what the f16 activation gradients can carry, so the gradient arena fills with inf / NaN.  Every
    such step must be SKIPPED -- weights, Adam moments and step count untouched, the scale halved -- until a scale fits and the
    step proceeds; the EMA update runs every step on whatever the weights are (train_realesrnet.py:394).  Flat-arena Adam (one
    Parameter, bench.py's default) and the per-tensor form."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import RealESRNetStep
    torch.manual_seed(4)
    g = R.Generator(3, 3, 4, n_blocks=2).cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    ema = R.EMA(g, 0.9)
    ema.register()
    opt = torch.optim.Adam([g.flat_parameter()] if flat else g.parameters(), 2e-4, (0.9, 0.99), fused=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40, growth_interval=10 ** 9)
    step = RealESRNetStep(g, ema, opt, scaler, None)
    gen = torch.Generator(device="cuda").manual_seed(2)
    lr = torch.rand(8, 3, 32, 32, device="cuda", generator=gen)
    hr = torch.rand(8, 3, 128, 128, device="cuda", generator=gen)
    w0 = g.flat_parameters().clone()
    skipped = 0
    for it in range(16):
        before, scale_before = g.flat_parameters().clone(), scaler.get_scale()
        shadow_before = ema._flat_shadow.clone()
        loss = step(hr, lr)
        after, scale_after = g.flat_parameters(), scaler.get_scale()
        assert torch.isfinite(after).all() and torch.isfinite(ema._flat_shadow).all(), it
        assert torch.equal(ema._flat_shadow, (1.0 - 0.9) * after + 0.9 * shadow_before) or \
            torch.allclose(ema._flat_shadow, 0.1 * after + 0.9 * shadow_before, rtol=1e-6, atol=1e-9), "EMA runs every step on the current weights"
        if torch.equal(before, after):
            skipped += 1
            assert scale_after == 0.5 * scale_before, (it, scale_before, scale_after)
            assert torch.equal(after, w0)
            states = [s for s in opt.state.values() if s]
            assert all(float(s["step"]) == 0 for s in states) and all(torch.isfinite(s["exp_avg"]).all() for s in states), "a skipped step must not touch Adam's state"
        else:
            assert scale_after == scale_before and torch.isfinite(loss), (it, scale_before, scale_after)
            break
    else:
        raise AssertionError("no step ever proceeded")
    assert skipped >= 2, skipped                     # the first scales did overflow
    states = [s for s in opt.state.values() if s]
    assert states and all(float(s["step"]) == 1 for s in states)
    assert not torch.equal(g.flat_parameters(), w0)
    R._lib.chain_health(sync=True)


def test_gradscaler_overflow_in_the_gan_step_shared_scaler():
    """The RealESRGAN step shares ONE GradScaler between the generator and the discriminator update and calls update() after
    each (reference train_realesrgan.py:485-487, 515-517): with a scale that overflows both, each half-step is skipped on its own
    and the scale halves once per skipped half; arenas (weights, spectral-norm u / v aside) stay untouched while skipped; the
    step recovers by itself."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import RealESRGANStep
    torch.manual_seed(6)
    g = R.Generator(3, 3, 4, n_blocks=1).cuda().train()
    d = R.Discriminator().cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    g_opt = torch.optim.Adam([g.flat_parameter()], 1e-4, (0.9, 0.99), fused=True)
    d_opt = torch.optim.Adam([d.flat_parameter()], 1e-4, (0.9, 0.99), fused=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 42, growth_interval=10 ** 9)
    step = RealESRGANStep(g, d, None, g_opt, d_opt, scaler, None)
    gen = torch.Generator(device="cuda").manual_seed(2)
    lr = torch.rand(8, 3, 16, 16, device="cuda", generator=gen)
    hr = torch.rand(8, 3, 64, 64, device="cuda", generator=gen)
    g0, d0 = g.flat_parameters().clone(), d.flat_parameters().clone()
    g_moved = d_moved = False
    halvings = 0
    g_steps = d_steps = 0
    for it in range(24):
        gb, db, sb = g.flat_parameters().clone(), d.flat_parameters().clone(), scaler.get_scale()
        out = step(hr, lr)
        ga, da, sa = g.flat_parameters(), d.flat_parameters(), scaler.get_scale()
        assert torch.isfinite(ga).all() and torch.isfinite(da).all(), it
        g_skip, d_skip = torch.equal(gb, ga), torch.equal(db, da)
        assert sa == sb * 0.5 ** (int(g_skip) + int(d_skip)), (it, sb, sa, g_skip, d_skip)
        halvings += int(g_skip) + int(d_skip)
        g_steps, d_steps = g_steps + int(not g_skip), d_steps + int(not d_skip)
        g_moved, d_moved = g_moved or not g_skip, d_moved or not d_skip
        if not g_skip and not d_skip:
            assert all(torch.isfinite(v) for v in out.values()), out
            break
    else:
        raise AssertionError("the step never recovered")
    assert halvings >= 3 and g_moved and d_moved
    assert not torch.equal(g.flat_parameters(), g0) and not torch.equal(d.flat_parameters(), d0)
    # Adam counted exactly the half-steps that were not skipped (the discriminator's gradients are larger: it may recover a step later)
    for opt, n_steps in ((g_opt, g_steps), (d_opt, d_steps)):
        states = [s for s in opt.state.values() if s]
        assert states and n_steps >= 1 and all(float(s["step"]) == n_steps for s in states), (n_steps, [float(s["step"]) for s in states])
