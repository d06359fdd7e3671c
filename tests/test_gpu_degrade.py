"""-m gpu parity of the degradation kernels (through the reference-named surface in
real_esrgan_pytorch_amd.imgproc, i.e. the C-ABI) against the golden vectors captured from the
reference and against the CPU oracle on injected random draws."""
import json
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k] for k in z.files}


@pytest.fixture(scope="module")
def ip():
    from real_esrgan_pytorch_amd import imgproc
    return imgproc


def err(a, b):
    return (a.cpu() - b).abs().max().item()


def test_usm_and_filter2d_vs_reference_golden(ip):
    g = load("imgproc_filter")
    x = g["x"].cuda()
    usm = ip.USMSharp(50, 0).cuda()
    assert torch.equal(usm.kernel.cpu(), g["usm_kernel"])
    assert err(usm(x, 0.5, 10), g["usm"]) < 2e-5          # separable 51-tap passes vs the dense 51x51 of the reference
    assert err(ip.filter2d_torch(x, g["k7"].cuda()), g["f7"]) < 1e-5
    assert err(ip.filter2d_torch(x, g["k21"].cuda()), g["f21"]) < 1e-5
    with pytest.raises(ValueError):
        ip.filter2d_torch(x, torch.ones(1, 4, 4).cuda())


@pytest.mark.parametrize("shape", [(2, 3, 72, 64), (1, 3, 400, 400), (3, 1, 31, 97), (1, 3, 26, 26), (2, 3, 130, 67)])
def test_usm_two_launches_equal_six_passes(ip, shape):
    """USMSharp(50, 0).forward as two fused launches (blur + byte mask; soft mask + combine, csrc/degrade.hip usm51_kernel) against
    the six separate passes (RESR_USM_SIX_PASSES=1): same taps in the same order -> the same values to an ulp, output AND the tensors
    the backward pass reads (blur, soft); ragged sizes, tiles cut by the image border, the smallest size the reflect padding allows."""
    from oracle import imgproc_ref as I
    n, c, h, w = shape
    gen = torch.Generator().manual_seed(h * w)
    import torch.nn.functional as F
    x = F.interpolate(torch.rand(n, c, max(2, h // 8), max(2, w // 8), generator=gen), size=(h, w), mode="bicubic").clamp(0, 1)
    x = (0.8 * x + 0.2 * torch.rand(n, c, h, w, generator=gen)).clamp(0, 1)
    usm = ip.USMSharp(50, 0).cuda()

    def run(six):
        if six:
            os.environ["RESR_USM_SIX_PASSES"] = "1"
        try:
            xd = x.cuda().requires_grad_(True)
            y = usm(xd, 0.5, 10)
            gw = torch.ones_like(y)
            y.backward(gw)
            torch.cuda.synchronize()
            return y.detach().clone(), xd.grad.clone()
        finally:
            os.environ.pop("RESR_USM_SIX_PASSES", None)
    y6, g6 = run(True)
    y2, g2 = run(False)
    # (one ulp apart where the compiler contracts the two kernels' multiply-adds differently; a flipped mask element would be ~1e-2)
    assert (y2 - y6).abs().max().item() <= 2.5e-7, (y2 - y6).abs().max().item()
    assert (g2 - g6).abs().max().item() <= 1e-6 * max(1.0, g6.abs().max().item()), (g2 - g6).abs().max().item()   # the backward pass reads the forward's blur / soft tensors
    # against the oracle's dense 51 x 51 blur: equal to rounding except where |x - blur| * 255 sits within rounding of the threshold --
    # a mask element that flips there moves its 51 x 51 neighbourhood of the soft mask by up to 4e-4 (the golden image of
    # test_usm_and_filter2d_vs_reference_golden has no such element; these random images may)
    ref = I.usm_sharp(x, I.usm_kernel(50, 0), 0.5, 10)
    d = (y2.cpu() - ref).abs()
    assert (d > 2e-5).float().mean().item() < 2e-2 and d.max().item() < 5e-3, ((d > 2e-5).float().mean().item(), d.max().item())
    frac = ((y2.cpu() - x).abs() > 1e-6).float().mean().item()
    assert frac > 0.01, "the mask never fired: the test image is too smooth to exercise the sharpening branch"


def test_resize_vs_reference_golden(ip):
    g = load("imgproc_resize")
    x = g["x"].cuda()
    for mode in ("area", "bilinear", "bicubic"):
        for s in (0.3731, 1.3177):
            ref = g[f"{mode}_sf_{s}"]
            got = ip.interpolate(x, scale_factor=s, mode=mode)
            assert got.shape == ref.shape, (mode, s)
            assert err(got, ref) < 2e-5, (mode, s)
        for tag, size in (("30x27", (30, 27)), ("12x10", (12, 10))):
            assert err(ip.interpolate(x, size=size, mode=mode), g[f"{mode}_size_{tag}"]) < 2e-5, (mode, tag)


def test_gaussian_noise_injected_draws_and_rng(ip):
    from oracle import imgproc_ref as I
    gen = torch.Generator().manual_seed(3)
    x = torch.round(torch.rand(4, 3, 24, 20, generator=gen) * 255) / 255
    sigma = torch.rand(4, generator=gen) * 29 + 1
    gray = torch.tensor([1.0, 0.0, 1.0, 0.0])
    fg, fc = torch.randn(24, 20, generator=gen), torch.randn(4, 3, 24, 20, generator=gen)
    ref = (x + I.gaussian_noise_from_fields(x, sigma, gray, fg, fc)).clamp(0, 1)
    got = ip.add_gaussian_noise_fields(x.cuda(), sigma.cuda(), gray.cuda(), fg.cuda(), fc.cuda(), True, False)
    assert err(got, ref) < 1e-6
    # gray samples share ONE field across the batch (reference quirk): noise of sample 0 and 2 is proportional
    n0, n2 = (got[0].cpu() - x[0]), (got[2].cpu() - x[2])
    g0, g2 = got[0].cpu(), got[2].cpu()
    inner = ((g0 > 0) & (g0 < 1) & (g2 > 0) & (g2 < 1)).all(dim=0)          # pixels not clipped in any channel
    assert inner.float().mean() > 0.3
    for ch in range(3):
        assert torch.allclose(n0[ch][inner] / sigma[0], n2[ch][inner] / sigma[2], atol=1e-5)
    assert torch.allclose(n0[0][inner], n0[1][inner], atol=1e-6)             # gray: same noise in every channel
    # the device RNG itself: moments of the Philox/Box-Muller field
    out = ip.random_add_gaussian_noise_torch(torch.full((8, 3, 128, 128), 0.5).cuda(), (10, 10), 0.0, False, False)
    z = (out.cpu() - 0.5) * 255 / 10
    assert abs(z.mean().item()) < 0.01 and abs(z.var().item() - 1) < 0.02 and abs((z ** 4).mean().item() - 3) < 0.1


def test_poisson_noise_vals_and_moments(ip):
    from oracle import imgproc_ref as I
    gen = torch.Generator().manual_seed(4)
    x = torch.round(torch.rand(3, 3, 40, 36, generator=gen) * 255) / 255
    x[1] = torch.round(x[1] * 7) / 7                      # few unique values -> small `vals`
    scale, gray = torch.tensor([1.0, 2.0, 0.5]), torch.tensor([0.0, 1.0, 0.0])
    out, vals = ip.add_poisson_noise(x.cuda(), scale.cuda(), gray.cuda(), 1234, True, False, return_vals=True)
    xq = torch.clamp((x * 255).round(), 0, 255) / 255
    gq = torch.clamp((I.rgb_to_gray(x) * 255).round(), 0, 255) / 255
    assert torch.equal(vals[:, 0].cpu(), I.poisson_vals(xq).flatten())          # imgproc.py:903-905, no host loop
    assert torch.equal(vals[:, 1].cpu(), I.poisson_vals(gq).flatten())          # imgproc.py:892-894
    # moments: noise = Poisson(q*v)/v - q has mean 0, variance q/v (before clipping); use a flat mid-gray image
    flat = torch.full((2, 3, 256, 256), 100 / 255.0)
    flat[:, :, :, ::2] = 120 / 255.0
    o2, v2 = ip.add_poisson_noise(flat.cuda(), torch.ones(2).cuda(), torch.zeros(2).cuda(), 99, False, False, return_vals=True)
    assert torch.equal(v2[:, 0].cpu(), torch.tensor([2.0, 2.0]))
    big = torch.round(torch.rand(1, 3, 256, 256, generator=gen) * 255) / 255
    o3, v3 = ip.add_poisson_noise(big.cuda(), torch.ones(1).cuda(), torch.zeros(1).cuda(), 7, False, False, return_vals=True)
    assert v3[0, 0].item() == 256.0
    n = (o3.cpu() - big)
    sel = (big > 0.3) & (big < 0.7)
    assert abs(n[sel].mean().item()) < 2e-4
    assert abs(n[sel].var().item() / (big[sel] / 256).mean().item() - 1) < 0.03
    lowsel = big < 0.02                                                         # lambda < 10: inversion branch
    assert abs(n[lowsel].mean().item()) < 2e-4
    assert abs(n[lowsel].var().item() / (big[lowsel] / 256).mean().item() - 1) < 0.1


def test_diff_jpeg_vs_reference_golden(ip, diag_dir):
    g = load("imgproc_jpeg")
    jpeg = ip.DiffJPEG(False)
    rep = {}
    for tag in ("48x40", "77x77", "100x100"):
        x, q = g[f"x_{tag}"].cuda(), g[f"q_{tag}"].cuda()
        y, c = jpeg(x, q, return_coeffs=True)
        ref_c = torch.cat([g[f"cy_{tag}"], g[f"ccb_{tag}"], g[f"ccr_{tag}"]], dim=1).reshape(c.shape)
        mism = (c.cpu() != ref_c).float().mean().item()
        e = (y.cpu() - g[f"y_{tag}"]).abs()
        rep[tag] = {"coef_mismatch_frac": mism, "max_err": e.max().item(), "p999": e.flatten().kthvalue(int(e.numel() * 0.999)).values.item()}
        assert mism < 2e-4, rep          # integer coefficients: identical except rounding ties at x.5
        assert rep[tag]["p999"] < 1e-4, rep
        assert torch.allclose(q.cpu(), g[f"factor_{tag}"], rtol=1e-6)       # the reference mutates its quality argument
    assert err(jpeg(g["x_48x40"].cuda(), 70), g["y_scalar_q70"]) < 1e-3
    with open(os.path.join(diag_dir, "jpeg_parity.json"), "w") as f:
        json.dump(rep, f)
    with pytest.raises(NotImplementedError):
        ip.DiffJPEG(True)


def test_quantize_crop_vs_reference_golden(ip):
    g = load("imgproc_crop")
    lrq = torch.clamp((g["lr"] * 255).round(), 0, 255) / 255
    plr, phr = ip.quantize_crop(g["lr"].cuda(), g["hr"].cuda(), 64, 4, int(g["top"]), int(g["left"]))
    ref_lr = lrq[:, :, int(g["top"]) // 4:int(g["top"]) // 4 + 16, int(g["left"]) // 4:int(g["left"]) // 4 + 16]
    assert torch.equal(plr.cpu(), ref_lr) and torch.equal(phr.cpu(), g["phr"])
    random.seed(5)
    a, b = ip.random_crop(g["lr"].cuda(), g["hr"].cuda(), 64, 4)
    assert torch.equal(a.cpu(), g["plr"]) and torch.equal(b.cpu(), g["phr"])


def test_pipeline_prefix_parity_and_prefetcher(ip):
    from oracle import imgproc_ref as I
    from real_esrgan_pytorch_amd import degrade
    random.seed(11)
    np.random.seed(11)
    torch.manual_seed(11)
    hr = torch.round(torch.rand(2, 3, 128, 160) * 255) / 255
    plan = degrade.sample_plan(2, 128, 160, 64)
    usm, jpeg = ip.USMSharp(50, 0).cuda(), ip.DiffJPEG(False)
    trace = {}
    lr, hrc = degrade.run_plan(hr.cuda(), plan, usm, jpeg, 4, 64, trace=trace)
    torch.cuda.synchronize()
    # deterministic prefix (before the first device-random draw) against the oracle, same plan
    ref = I.usm_sharp(hr, I.usm_kernel(50, 0))
    assert err(trace["usm"], ref) < 2e-5
    if plan.blur1:
        ref = I.filter2d(ref, torch.from_numpy(plan.kernel1))
        assert err(trace["blur1"], ref) < 2e-5
    ref = torch.nn.functional.interpolate(ref, scale_factor=plan.resize1_scale, mode=plan.resize1_mode)
    assert trace["resize1"].shape == ref.shape and err(trace["resize1"], ref) < 5e-5
    assert lr.shape == (2, 3, 16, 16) and hrc.shape == (2, 3, 64, 64)
    assert torch.equal(hrc.cpu(), hr[:, :, plan.hr_top:plan.hr_top + 64, plan.hr_left:plan.hr_left + 64])   # un-sharpened HR target
    v = lr.cpu() * 255
    assert (v - v.round()).abs().max().item() < 1e-4 and lr.min() >= 0 and lr.max() <= 1
    assert trace["final"].shape[2:] == (32, 40)
    # prefetching stage: results come one submission late, shapes stable, side stream used
    d = degrade.Degrader(batch=2, hr_size=128, upscale=4, crop=64, seed=1)
    hr2 = torch.round(torch.rand(2, 3, 128, 128) * 255).cuda() / 255
    for _ in range(3):
        a, b = d(hr2)
        assert a.shape == (2, 3, 16, 16) and b.shape == (2, 3, 64, 64)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()


def test_generic_filter_kernel_knob():
    """21x21 and 1x51 / 51x1 filters have register-tiled kernels; the generic kernel they replace stays correct for those sizes
    too: re-run the golden comparison in a subprocess with RESR_FILTER_GENERIC=1 (the knob is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RESR_FILTER_GENERIC="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_degrade.py"), "-q", "-x", "-m", "gpu",
                        "-k", "usm_and_filter2d or pipeline_prefix"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_degrader_pair_does_not_alias_a_buffer_the_caller_refills():
    """with an identity HR window (crop = the tile's edge: every batch of the reference's configuration) quantize_crop
    hands back the input tensor itself, and the prefetching stage returns its pairs ONE CALL LATER -- a caller that refills its HR
    buffer in place between two calls would get LR(i) next to HR(i + 1).  The stage keeps its own copy: the HR tensor of pair i still
    holds batch i's values after the buffer has been overwritten, and it is not the caller's storage."""
    from real_esrgan_pytorch_amd.degrade import Degrader
    d = Degrader(batch=2, hr_size=64, upscale=4, crop=64, seed=0)
    buf = torch.rand(2, 3, 64, 64, device="cuda")
    first = buf.clone()
    d(buf)                                   # submits batch 0 (and returns it: the very first call has nothing older)
    torch.cuda.synchronize()
    buf.fill_(0.25)                          # the caller refills its buffer in place: batch 1
    lr, hr = d(buf)                          # the pair submitted for batch 0 ... 
    torch.cuda.synchronize()
    assert hr.data_ptr() != buf.data_ptr()
    assert torch.equal(hr, first), "the HR half of a prefetched pair changed with the caller's buffer"
    assert lr.shape == (2, 3, 16, 16)
