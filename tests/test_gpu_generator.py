"""-m gpu parity of the whole generator (forward + backward) against the CPU oracle."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(upscale, n_blocks, seed, precision, x2_plan=0):
    """exact16 here is the all-pairs plan (x2_plan = 0: three stages per chunk everywhere); the single-f16 growth planes /
    growth-plane gradients that `Generator(precision="exact16")` takes by default have their own gates in test_gpu_x2_plan.py."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_generator_state(seed, 3, 3, upscale, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < n_blocks}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5     # centre the output inside the clamp range
    g = R.Generator(3, 3, upscale, precision=precision, n_blocks=n_blocks, x2_plan=x2_plan)
    g.load_state_dict(sd)
    return g.cuda(), sd, M


CASES = [(4, 23, 1, 24, 24), (4, 2, 1, 20, 24), (4, 1, 2, 33, 17), (2, 1, 1, 24, 40), (1, 1, 1, 16, 32)]
_ORACLE_CACHE = {}


@pytest.mark.parametrize("precision", ["strict", "exact16", "fast"])
@pytest.mark.parametrize("upscale,n_blocks,n,h,w", CASES)
def test_generator_forward_backward(upscale, n_blocks, n, h, w, precision, diag_dir):
    g, sd, M = _setup(upscale, n_blocks, 11, precision)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen)
    gw = torch.randn(n, 3, h * upscale, w * upscale, generator=gen)

    # oracle: the CPU restatement under fp32 autograd (the reference's own arithmetic) and the same code in float64.
    # LeakyReLU'(v) jumps 0.2 -> 1 at v = 0, so a pre-activation within rounding of zero flips one mask element between
    # two correct evaluations: on the 23-block case below the fp32 CPU path itself sits 1.8e-3 (relative L2, worst
    # tensor) from its float64 evaluation because one u2 pre-activation is 6.6e-8.  Gradient parity is therefore
    # measured against the float64 evaluation; the distance to the fp32 one is reported next to the fp32 path's own.
    def run_oracle(dt):
        sdo = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.to(dt).clone().requires_grad_(True)
        yo = M.generator_forward(xo, sdo, upscale, n_blocks)
        (yo * gw.to(dt)).sum().backward()
        return yo.detach(), {k: v.grad for k, v in sdo.items()}, xo.grad
    key = (upscale, n_blocks, n, h, w)
    if key not in _ORACLE_CACHE:      # (one pair of CPU evaluations per case: the three precisions share it -- 20 s each at 23 blocks, side by side)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as ex:
            f32, f64 = ex.submit(run_oracle, torch.float32), ex.submit(run_oracle, torch.float64)
            _ORACLE_CACHE[key] = (f32.result(), f64.result())
    (yo, go32, gxo32), (yo64, go64, gxo64) = _ORACLE_CACHE[key]

    # fast / exact16 keep activation gradients in f16 (pairs): scale the loss like the reference's GradScaler
    # (train_realesrnet.py:388) so they stay in the normal range, then unscale
    loss_scale = 1.0 if precision == "strict" else 1024.0
    xd = x.cuda().requires_grad_(True)
    y = g(xd)
    (y * gw.cuda()).sum().mul(loss_scale).backward()
    torch.cuda.synchronize()

    # north_star: 1e-3 max-abs vs the CPU path -- met by strict (f32 MFMA) and exact16 (split-operand f16 MFMA)
    # fast mode is held to 1.5 x what it measures on each case (the arithmetic is deterministic), not to a class bound a 2 x regression
    # would still pass under: forward 2.37e-3 at 23 blocks, 3.7-5.6e-5 on the shallow cases
    tol_y = (3.6e-3 if n_blocks == 23 else 1e-4) if precision == "fast" else 1e-3
    err_y = (y.detach().cpu() - yo).abs().max().item()
    rep = {"err_y": err_y, "err_y_vs_f64": (y.detach().cpu().double() - yo64).abs().max().item(),
           "frac_unclamped": ((yo > 0) & (yo < 1)).float().mean().item()}

    def rel_l2(got, ref):   # relative L2 per tensor
        return ((got.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-12)).item()

    worst = worst32 = own32 = 0.0
    for name, p in g.named_parameters():
        got = p.grad.cpu() / loss_scale
        e = rel_l2(got, go64[name])
        rep["g_" + name] = e
        worst = max(worst, e)
        worst32 = max(worst32, rel_l2(got, go32[name]))
        own32 = max(own32, rel_l2(go32[name], go64[name]))
    egx = rel_l2(xd.grad.cpu() / loss_scale, gxo64)
    rep.update(gx=egx, worst_vs_f64=worst, worst_vs_f32_oracle=worst32, f32_oracle_own_vs_f64=own32)
    with open(os.path.join(diag_dir, f"gen_{precision}_{upscale}_{n_blocks}_{n}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert err_y < tol_y, f"forward max abs err {err_y}"
    if precision == "strict":
        assert err_y < 5e-5, f"strict forward err {err_y}"
    if precision == "exact16":
        assert err_y < 2e-4, f"exact16 forward err {err_y}"
    # fast: same class as the reference's own CUDA-autocast path -- torch CPU autocast(fp16) of the oracle vs the
    # fp32 oracle measures err_y 2.3e-3 and worst per-tensor rel-L2 4-5 % (with infs) on the 23-block case.
    # strict follows the fp32 path's roundings (and its mask flips); exact16 is checked against the float64 evaluation.
    if precision == "fast":
        # measured (worst tensor, input gradient) per case under this dense random cotangent -- mask flips of an f16 forward included:
        # x1: 5.1e-2 / 1.4e-2, x2: 3.5e-2 / 2.3e-2, x4 1 block: 3.0e-2 / 2.7e-2, x4 2 blocks: 5.8e-2 / 2.7e-2, x4 23 blocks: 8.3e-2 / 4.6e-2
        gate_w, gate_gx = {(1, 1): (0.077, 0.021), (2, 1): (0.053, 0.034), (4, 1): (0.046, 0.041), (4, 2): (0.087, 0.041), (4, 23): (0.125, 0.069)}[(upscale, n_blocks)]
        assert worst < gate_w and egx < gate_gx, f"worst rel grad err {worst} (gate {gate_w}), gx {egx} (gate {gate_gx})"
    elif precision == "strict":
        assert worst32 < 1e-2, f"worst rel grad err vs the fp32 oracle {worst32}"
    else:
        # every gradient tensor (702 at full depth) and the input gradient: the gate is 1e-3 relative L2; with the default
        # three-product weight gradients the measured worst tensor is 6.5e-6 over the five cases -- held to 5e-5 so that a
        # regression to the hi-only class (3-9e-4, still inside the gate) cannot pass unnoticed
        assert worst < 5e-5 and egx < 1e-5, f"exact16 worst rel grad err vs f64 {worst}, gx {egx} (fp32 oracle's own: {own32})"
        assert worst32 < max(1e-3, 1.5 * own32), f"exact16 vs the fp32 oracle {worst32} (its own distance to f64: {own32})"


@pytest.mark.parametrize("precision,upscale", [("strict", 4), ("exact16", 4), ("exact16", 2), ("fast", 4)])
def test_generator_inference_matches_training_forward(precision, upscale):
    g, sd, M = _setup(upscale, 3, 3, precision)
    x = torch.rand(2, 3, 40, 36).cuda()
    with torch.no_grad():
        y0 = g(x)                      # rotating-workspace inference plan
    y1 = g(x.clone().requires_grad_(True))   # saved-activation training plan
    assert torch.equal(y0, y1.detach())
    yo = M.generator_forward(x.cpu(), sd, upscale, 3)
    assert (y0.cpu() - yo).abs().max().item() < (5e-3 if precision == "fast" else 1e-4)


def test_state_dict_surface_and_channels_last():
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="strict", n_blocks=1)
    keys = list(g.state_dict().keys())
    assert keys[:2] == ["conv1.weight", "conv1.bias"] and "trunk.0.rdb3.conv5.bias" in keys
    assert "upsampling1.0.weight" in keys and "conv3.0.bias" in keys and keys[-1] == "conv4.bias"
    g = g.to(memory_format=torch.channels_last, device="cuda")     # reference inference.py:28
    x = torch.rand(1, 3, 16, 16).cuda()
    with torch.no_grad():
        a = g(x)
        b = g(x.to(memory_format=torch.channels_last))              # reference inference.py:49
    assert torch.equal(a, b) and a.shape == (1, 3, 64, 64)
    with pytest.raises(RuntimeError):
        g(x.cpu())


@pytest.mark.parametrize("precision,size,plan", [(None, (24, 28), None), ("strict", (24, 28), None), (None, (128, 128), None), (None, (24, 28), "59")])
def test_inference_entry_point(tmp_path, monkeypatch, precision, size, plan):
    """reference inference.py flow: checkpoint with 'model.'-prefixed keys -> PNG in -> PNG out, vs the oracle.
    precision None = the entry point's DEFAULT (no --precision): the reference runs this call site in fp32
    (inference.py:52-53, no autocast), so the default must be a mode inside the 1e-3 tolerance (config.inference_precision).
    The default exact16 plan carries the MX-fp8 correction stages since round 6 (forward ~1e-4 instead of 2e-6): an image then differs
    from the oracle's by one uint8 level on ~0.3 % of its values (a truncating conversion crosses a level wherever the value sits
    within the error of one); $RESR_X2_PLAN=59 (three f16 stages per pair chunk) keeps round 5's < 0.1 %."""
    if plan is not None:
        monkeypatch.setenv("RESR_X2_PLAN", plan)
    import numpy as np
    from PIL import Image
    from oracle import model_ref as M
    from real_esrgan_pytorch_amd import inference
    sd = M.init_generator_state(21, bias_noise=0.02)
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, tmp_path / "g.pth.tar")
    rng = np.random.RandomState(0)
    lr = rng.randint(0, 256, size=(size[0], size[1], 3), dtype=np.uint8)      # (128, 128): BASELINE config 1's geometry
    Image.fromarray(lr).save(tmp_path / "lr.png")

    class A:
        inputs_path, output_path, weights_path = str(tmp_path / "lr.png"), str(tmp_path / "sr.png"), str(tmp_path / "g.pth.tar")
    if precision is not None:
        A.precision = precision
    else:
        from real_esrgan_pytorch_amd import config
        assert config.inference_precision == "exact16" and config.precision == "fast"   # fp32 call sites / autocast call sites
    inference.main(A)
    got = np.asarray(Image.open(tmp_path / "sr.png")).astype(np.int32)
    x = torch.from_numpy(lr.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
    ref = M.generator_forward(x, sd, 4).squeeze(0).permute(1, 2, 0).mul(255).clamp(0, 255).numpy().astype("uint8").astype(np.int32)
    assert got.shape == (4 * size[0], 4 * size[1], 3)
    d = np.abs(got - ref)
    lim = 1e-2 if (precision is None and plan is None) else 1e-3
    assert d.max() <= 1 and (d > 0).mean() < lim, (d.max(), (d > 0).mean())      # truncating uint8 conversion: a value within the error of a level may flip it


def test_directory_test_entry_point_default_precision(tmp_path, monkeypatch):
    """reference test.py flow on its DEFAULT precision (test.py:79-80 runs fp32, no autocast): EMA weights from a checkpoint,
    every LR PNG of a folder -> SR PNG + NIQE.  The written images equal the oracle's uint8 images (<= 1 LSB on < 1 % of the
    values: the default plan's MX stages, ~1e-4) and the reported NIQE equals the NIQE of the oracle's SR tensors."""
    import numpy as np
    from PIL import Image
    from oracle import model_ref as M
    from real_esrgan_pytorch_amd import config, image_quality_assessment as IQA
    from real_esrgan_pytorch_amd import test as E
    sd = M.init_generator_state(23, bias_noise=0.02)
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    torch.save({"ema_state_dict": {"model." + k: v for k, v in sd.items()}}, tmp_path / "g.pth.tar")
    os.makedirs(tmp_path / "lr")
    import torch.nn.functional as F
    lrs = {}
    for i, (name, (h, w)) in enumerate((("a.png", (56, 56)), ("b.png", (48, 64)))):
        # smooth + grain: NIQE wants image-like statistics and several 96 x 96 blocks of the x4 output
        g = torch.Generator().manual_seed(50 + i)
        x = F.interpolate(torch.rand(1, 3, h // 8, w // 8, generator=g), size=(h, w), mode="bicubic").clamp(0, 1)
        x = (0.85 * x + 0.15 * torch.rand(1, 3, h, w, generator=g)).clamp(0, 1)
        lrs[name] = (x[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
        Image.fromarray(lrs[name]).save(tmp_path / "lr" / name)
    here = os.path.dirname(os.path.abspath(__file__))
    for k, v in dict(lr_dir=str(tmp_path / "lr"), sr_dir=str(tmp_path / "sr"), model_path=str(tmp_path / "g.pth.tar"),
                     niqe_model_path=os.path.join(here, "golden", "niqe_model.mat"), device=torch.device("cuda", 0)).items():
        monkeypatch.setattr(config, k, v, raising=False)
    assert config.inference_precision == "exact16"
    score = E.main()
    niqe = IQA.NIQE(4, config.niqe_model_path).cuda()
    ref_scores = []
    for name, lr in lrs.items():
        x = torch.from_numpy(lr.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        yo = M.generator_forward(x, sd, 4)
        ref = yo.squeeze(0).permute(1, 2, 0).mul(255).clamp(0, 255).numpy().astype("uint8").astype(np.int32)
        got = np.asarray(Image.open(tmp_path / "sr" / name)).astype(np.int32)
        d = np.abs(got - ref)
        assert got.shape == ref.shape and d.max() <= 1 and (d > 0).mean() < 1e-2, (name, d.max(), (d > 0).mean())    # (MX stages: see test_inference_entry_point)
        ref_scores.append(niqe(yo.cuda()).item())
    assert score == score and 0 < score <= 100, score
    assert abs(score - sum(ref_scores) / len(ref_scores)) < 3e-3 * max(1.0, abs(score)), (score, ref_scores)


def test_entry_points_route_large_frames_through_the_tiler(tmp_path):
    """A frame beyond the conv kernels' 2^24-pixel tensors (here 1100 x 1000 LR -> 17.6 M HR pixels) used to raise in
    inference.main; tiling.super_resolve cuts it into haloed tiles.  With a 1-block trunk (receptive field ~21 LR px < the
    default halo) the stitched image equals what two half-frames with a generous overlap give where both cover a pixel; a frame
    that fits runs as one pass, bit-equal to model(x)."""
    import numpy as np
    from PIL import Image
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd import inference, tiling
    torch.manual_seed(3)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=1).cuda().eval()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    small = torch.rand(1, 3, 40, 56, device="cuda")
    assert tiling.fits_whole(g, 1, 40, 56) and torch.equal(tiling.super_resolve(g, small), g(small).detach())
    H, W = 1100, 1000
    assert not tiling.fits_whole(g, 1, H, W)
    x = torch.rand(1, 3, H, W, device="cuda")
    y = tiling.super_resolve(g, x)
    assert y.shape == (1, 3, 4 * H, 4 * W) and torch.isfinite(y).all()
    with torch.no_grad():
        top = g(x[:, :, :640].contiguous())            # fits: 640 x 1000 x 16 = 10.2 M HR pixels
        bot = g(x[:, :, H - 640:].contiguous())
    assert torch.equal(y[:, :, :4 * 500], top[:, :, :4 * 500])                 # rows far from the cut of either half
    assert torch.equal(y[:, :, 4 * 600:], bot[:, :, 4 * (600 - (H - 640)):])
    # ... and through the CLI entry point (default precision = exact16): no exception, the file has the frame's size
    torch.save({"state_dict": {"model." + k: v for k, v in g.state_dict().items()}}, tmp_path / "g.pth.tar")
    Image.fromarray((x[0, :, :1050, :1000].permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)).save(tmp_path / "lr.png")

    class A:
        inputs_path, output_path, weights_path, precision = str(tmp_path / "lr.png"), str(tmp_path / "sr.png"), str(tmp_path / "g.pth.tar"), "fast"
    from real_esrgan_pytorch_amd import model as model_mod
    orig = model_mod.Generator.N_BLOCKS
    model_mod.Generator.N_BLOCKS = 1
    try:
        inference.main(A)
    finally:
        model_mod.Generator.N_BLOCKS = orig
    assert Image.open(tmp_path / "sr.png").size == (4000, 4200)


@pytest.mark.parametrize("use_graph", [False, True])
def test_tiled_inference_equals_whole_image(use_graph):
    """Halo >= receptive-field radius (1-block trunk: ~21 LR px)  =>  stitched tiles == whole-image pass, bit for bit."""
    from real_esrgan_pytorch_amd.tiling import TiledGenerator
    g, sd, M = _setup(2, 1, 5, "fast")
    x = torch.rand(1, 3, 144, 176).cuda()
    with torch.no_grad():
        whole = g(x)
    tiled = TiledGenerator(g, tile=64, halo=24, use_graph=use_graph)(x)
    assert tiled.shape == whole.shape == (1, 3, 288, 352)
    assert torch.equal(tiled, whole)
    # rectangular tiles; the captured whole-frame graph replayed on a second frame; re-capture after the parameters moved
    tg = TiledGenerator(g, tile=(48, 80), halo=24, use_graph=use_graph)
    assert torch.equal(tg(x), whole)
    x2 = torch.rand(1, 3, 144, 176).cuda()
    with torch.no_grad():
        whole2 = g(x2)
    assert torch.equal(tg(x2), whole2)
    with torch.no_grad():
        for p in g.parameters():                     # what EMA.apply_shadow does: new storage behind every parameter
            p.data = (p.data * 1.01).clone()
        whole3 = g(x2)
    assert not torch.equal(whole3, whole2)
    assert torch.equal(tg(x2), whole3)


@pytest.mark.parametrize("precision", ["strict", "exact16", "fast"])
def test_dense_blocks_standalone_forward(precision):
    """ResidualDenseBlock / ResidualResidualDenseBlock are public surface of the reference (model.py:22-27): their standalone
    forward against the reference's own outputs (tests/golden/rdb.npz, rrdb.npz)."""
    import numpy as np
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.model import _dense_blocks_forward
    here = os.path.dirname(os.path.abspath(__file__))
    tol = 2e-2 if precision == "fast" else 2e-5
    z = np.load(os.path.join(here, "golden", "rdb.npz"))
    rdb = R.ResidualDenseBlock(64, 32)
    rdb.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_")})
    rdb = rdb.cuda()
    with torch.no_grad():
        y = _dense_blocks_forward([rdb], torch.from_numpy(z["x"]).cuda(), False, precision).cpu()
    assert (y - torch.from_numpy(z["y"])).abs().max().item() < tol
    with pytest.raises(RuntimeError):          # forward only: autograd must go through the Generator
        rdb(torch.from_numpy(z["x"]).cuda().requires_grad_(True))
    z = np.load(os.path.join(here, "golden", "rrdb.npz"))
    rrdb = R.ResidualResidualDenseBlock(64, 32)
    rrdb.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith("w_")})
    rrdb = rrdb.cuda()
    with torch.no_grad():
        y = _dense_blocks_forward([rrdb.rdb1, rrdb.rdb2, rrdb.rdb3], torch.from_numpy(z["x"]).cuda(), True, precision).cpu()
        y_env = rrdb(torch.from_numpy(z["x"]).cuda()).cpu()          # module surface: precision from $RESR_PRECISION (fast)
    assert (y - torch.from_numpy(z["y"])).abs().max().item() < tol
    assert (y_env - torch.from_numpy(z["y"])).abs().max().item() < 2e-2


def test_two_backwards_accumulate():
    """Two backward passes without zero_grad in between must leave the SUM in .grad (the native backward overwrites its
    gradient arena; model.py parks and restores earlier gradients)."""
    g, sd, M = _setup(4, 1, 3, "strict")
    gen = torch.Generator().manual_seed(3)
    xa, xb = torch.rand(1, 3, 16, 16, generator=gen), torch.rand(1, 3, 16, 16, generator=gen)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (M.generator_forward(xa, sdo, 4, 1).square().sum() + M.generator_forward(xb, sdo, 4, 1).square().sum()).backward()
    g(xa.cuda()).square().sum().backward()
    g(xb.cuda()).square().sum().backward()
    torch.cuda.synchronize()
    for name, p in g.named_parameters():
        ref = sdo[name].grad
        assert ((p.grad.cpu() - ref).norm() / ref.norm().clamp_min(1e-12)).item() < 1e-4, name
    # two live graphs before either backward (GAN-style): each keeps its own workspace
    g.zero_grad(set_to_none=True)
    ya, yb = g(xa.cuda()), g(xb.cuda())
    (ya.square().sum() + yb.square().sum()).backward()
    torch.cuda.synchronize()
    for name, p in g.named_parameters():
        ref = sdo[name].grad
        assert ((p.grad.cpu() - ref).norm() / ref.norm().clamp_min(1e-12)).item() < 1e-4, name


def test_exact16_three_product_weight_gradients_knob():
    """exact16's weight gradients use all three tap-products by default (X_hi^T G_hi + 2^-12 (X_hi^T G_lo + X_lo^T G_hi)): every
    one of the 702 tensors of the 23-block case within 2e-5 of the float64 evaluation (measured 5.8e-6).  RESR_X2_WGRAD_PRODUCTS=1
    is the opt-in hi-tensors-only form (a third of the matrix work, 3-9e-4 on 24^2 ... 2 x 256^2,
    profiles/r03_x2_wgrad_validate.json: inside 1e-3 without real margin, hence not the default).  The knob is read per call."""
    g, sd, M = _setup(4, 23, 11, "exact16")
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 24, 24, generator=gen)
    gw = torch.randn(1, 3, 96, 96, generator=gen)
    sdo = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    (M.generator_forward(x.double(), sdo, 4, 23) * gw.double()).sum().backward()
    worst = {}
    for products in ("1", "3", ""):
        os.environ["RESR_X2_WGRAD_PRODUCTS"] = products
        try:
            g.zero_grad(set_to_none=True)
            (g(x.cuda()) * gw.cuda()).sum().mul(1024.0).backward()
            torch.cuda.synchronize()
        finally:
            os.environ.pop("RESR_X2_WGRAD_PRODUCTS", None)
        worst[products] = max(((p.grad.cpu().double() / 1024.0 - sdo[n].grad).norm() / sdo[n].grad.norm().clamp_min(1e-12)).item()
                              for n, p in g.named_parameters())
    assert worst["3"] < 2e-5, worst
    assert worst[""] == worst["3"], worst          # the default IS the three-product form
    assert worst["3"] < worst["1"] < 1e-3, worst


def test_fast_mode_gradients_under_the_train_loss_stay_in_their_class(diag_dir):
    """What training consumes (bench.py `parity_mode.gradient_probe`, here at 4 x 64^2): every gradient tensor of the benchmarked f16 mode and of
    exact16's default plan against exact16's all-pairs plan under the train step's own L1 mean loss at a GradScaler's initial scale.
    The driver line of round 5 read median 9.9e-4 / worst 2.3e-3 for fast mode at 16 x 256^2 and 2e-5 for exact16's default plan: a test
    now holds fast mode to 3e-3 and exact16's default plan -- MX backward-data stages since round 6 -- to 5e-4."""
    import bench
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast")
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    rec = bench.gradient_probe(sd, 4, 64)
    with open(os.path.join(diag_dir, "gradient_probe_4x64.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert rec["fast_f16"]["worst"] <= 3e-3 and rec["fast_f16"]["median"] <= 1.5e-3, rec
    assert rec["exact16_default_plan"]["worst"] <= 5e-4, rec
