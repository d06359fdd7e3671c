"""-m gpu parity of the spectral-norm U-Net discriminator against the CPU oracle and the reference golden."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _make(precision, seed):
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_discriminator_state(seed)
    d = R.Discriminator(precision=precision)
    d.load_state_dict(sd)
    return d.cuda(), sd, M


@pytest.mark.parametrize("golden", ["discriminator", "discriminator_flipfree"])
@pytest.mark.parametrize("precision", ["strict", "exact16", "fast"])
def test_three_training_calls_vs_reference_golden(precision, golden, diag_dir):
    z = np.load(os.path.join(G, golden + ".npz"))
    g = {k: torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k] for k in z.files}
    d, sd, M = _make(precision, int(g["seed"]))
    d.train()
    x = g["x"].cuda().requires_grad_(True)
    scale = 1.0 if precision == "strict" else 256.0
    # "discriminator" (seed 201): one LeakyReLU pre-activation of the 8x8 level lies within fp32 rounding of zero; two correct
    # evaluations differ by that mask element (~1e-2 in the affected gradients): tolerance 2e-2.  "discriminator_flipfree"
    # (seed 231, tests/golden/gen_golden.py): no such element -- strict AND exact16 (hi/lo f16 pairs on the f16 matrix pipe, the
    # mode that meets north_star's 1e-3) are held to 1e-3 on every checked gradient.
    tol_y, tol_g = (2e-4, 1e-3 if golden == "discriminator_flipfree" else 2e-2) if precision != "fast" else (2e-2, 0.12)
    rep = {}
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    for call in range(3):                      # train_realesrgan.py:479,500,508: three training-mode forwards per step
        d.zero_grad()
        x.grad = None
        y = d(x)
        (y * g["gw"].cuda()).sum().mul(scale).backward()
        torch.cuda.synchronize()
        named = dict(d.named_parameters())
        rep[f"call{call}"] = {
            "y": (y.detach().cpu() - g[f"y{call}"]).abs().max().item(),
            "gx": rel(x.grad.cpu() / scale, g[f"gx{call}"]),
            "u": (d.state_dict()["up_block1.0.weight_u"].cpu() - g[f"u{call}_up_block1"]).abs().max().item(),
            "v": (d.state_dict()["down_block3.0.weight_v"].cpu() - g[f"v{call}_down_block3"]).abs().max().item(),
            "g_conv1": rel(named["conv1.weight"].grad.cpu() / scale, g[f"g{call}_conv1.weight"]),
            "g_down2": rel(named["down_block2.0.weight_orig"].grad.cpu()[:4] / scale, g[f"g{call}_down_block2.weight_orig"]),
            "g_conv3": rel(named["conv3.0.weight_orig"].grad.cpu()[:8] / scale, g[f"g{call}_conv3.weight_orig"]),
            "g_conv4b": rel(named["conv4.bias"].grad.cpu() / scale, g[f"g{call}_conv4.bias"]),
        }
    with open(os.path.join(diag_dir, f"disc_{precision}{'_flipfree' if golden != 'discriminator' else ''}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    for call in range(3):
        r = rep[f"call{call}"]
        assert r["y"] < tol_y * max(1.0, g[f"y{call}"].abs().max().item()), rep
        assert r["u"] < 1e-5 and r["v"] < 1e-5, rep             # power iteration is fp32 in both modes
        for k in ("gx", "g_conv1", "g_down2", "g_conv3", "g_conv4b"):
            assert r[k] < tol_g, (k, rep)
    d.eval()
    with torch.no_grad():
        ye = d(g["x"].cuda())
    assert (ye.cpu() - g["y_eval"]).abs().max().item() < tol_y * max(1.0, g["y_eval"].abs().max().item())


@pytest.mark.parametrize("golden", ["discriminator_allgrads_240", "discriminator_allgrads_243"])
@pytest.mark.parametrize("precision", ["strict", "exact16", "fast"])
def test_every_gradient_tensor_vs_reference_golden(precision, golden, diag_dir):
    """The 1e-3 gate of strict / exact16 on EVERY gradient tensor of the three training calls, not a slice of four, and on seeds
    that were not searched for.  What the two seeds show (gpurun_out/disc_allgrads_*.json): both are free of
    near-zero LeakyReLU pre-activations in the fp32 CPU evaluation (oracle fp32 vs float64 1.2-1.5e-6) -- but "flip-free" is a
    property of ONE evaluation's roundings.  Seed 240: every tensor of every call within 1e-3 on the GPU too (strict 4e-4,
    exact16 4e-4).  Seed 243: the GPU's roundings put another pre-activation across zero in one call (strict: call 0 below
    up_block2; exact16: call 1 below conv3): every tensor BELOW that layer moves by 1e-4 .. 5e-3, every tensor above it stays at
    1e-6, the forward output at 3e-7 -- the signature of one flipped mask element, which no arithmetic of finite precision is
    immune to (the fp32 reference flips on seed 201).  Gates: 1e-3 on seed 240; on seed 243 1e-2 per tensor, 1e-5 on the tensors
    above every LeakyReLU (conv4), and at most one call with a tensor beyond 1e-3."""
    z = np.load(os.path.join(G, golden + ".npz"))
    g = {k: torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k] for k in z.files}
    d, sd, M = _make(precision, int(g["seed"]))
    d.train()
    names = [str(n) for n in g["names"]]
    x = g["x"].cuda().requires_grad_(True)
    scale = 1.0 if precision == "strict" else 256.0
    tol_y, tol_g = (2e-4, 1e-3) if precision != "fast" else (2e-2, 0.12)

    def sub(t, cap=4096):
        f = t.reshape(-1)
        return f if f.numel() <= cap else f[::(f.numel() + cap - 1) // cap]
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    rep = {}
    for call in range(3):
        d.zero_grad()
        x.grad = None
        y = d(x)
        (y * g["gw"].cuda()).sum().mul(scale).backward()
        torch.cuda.synchronize()
        named = dict(d.named_parameters())
        assert list(named) == names
        r = {"y": (y.detach().cpu() - g[f"y{call}"]).abs().max().item(), "gx": rel(x.grad.cpu() / scale, g[f"gx{call}"])}
        for i, k in enumerate(names):
            got = named[k].grad.cpu() / scale
            r[k] = rel(sub(got), g[f"g{call}_{i}"])
            r["norm_" + k] = abs(got.double().norm().item() / float(g[f"n{call}_{i}"]) - 1.0)
        rep[f"call{call}"] = r
    with open(os.path.join(diag_dir, f"disc_allgrads_{precision}_{golden[-3:]}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    flip_tolerant = golden.endswith("243") and precision != "fast"
    gate = 1e-2 if flip_tolerant else tol_g
    calls_beyond = 0
    for call in range(3):
        r = rep[f"call{call}"]
        assert r["y"] < tol_y * max(1.0, g[f"y{call}"].abs().max().item()) and r["gx"] < gate, (call, r)
        for k in names:
            assert r[k] < gate and r["norm_" + k] < gate, (call, k, r[k], r["norm_" + k])
        if precision != "fast":
            assert r["conv4.weight"] < 1e-5 and r["conv4.bias"] < 1e-5, (call, r)      # nothing between them and the loss can flip
        calls_beyond += int(max(r[k] for k in names) > tol_g)
    assert calls_beyond <= (1 if flip_tolerant else 0), rep


@pytest.mark.parametrize("precision", ["strict", "exact16"])
def test_all_gradients_vs_oracle_odd_shape(precision):
    d, sd, M = _make(precision, 7)
    scale = 1.0 if precision == "strict" else 256.0
    d.train()
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(1, 3, 40, 72, generator=gen)
    gw = torch.randn(1, 1, 40, 72, generator=gen)
    sdo = {k: v.clone() for k, v in sd.items()}
    for k in sdo:
        if not (k.endswith("_u") or k.endswith("_v")):
            sdo[k].requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    (M.discriminator_forward(xo, sdo, True) * gw).sum().backward()
    xd = x.cuda().requires_grad_(True)
    (d(xd) * gw.cuda()).sum().mul(scale).backward()
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    for name, p in d.named_parameters():
        assert rel(p.grad.cpu() / scale, sdo[name].grad) < 5e-3, name
    assert rel(xd.grad.cpu() / scale, xo.grad) < 5e-3
    # frozen discriminator (generator step, train_realesrgan.py:466-467): only the input gradient flows
    d.zero_grad(set_to_none=True)
    for p in d.parameters():
        p.requires_grad = False
    for v in sdo.values():
        v.grad = None
    xo2 = x.clone().requires_grad_(True)
    (M.discriminator_forward(xo2, sdo, True) * gw).sum().backward()      # (the oracle's u / v have advanced one call, like the module's)
    xd2 = x.cuda().requires_grad_(True)
    (d(xd2) * gw.cuda()).sum().mul(scale).backward()
    assert all(p.grad is None for p in d.parameters()), "a frozen discriminator must not receive parameter gradients"
    assert rel(xd2.grad.cpu() / scale, xo2.grad) < 5e-3
    with pytest.raises(RuntimeError):
        d(torch.rand(1, 3, 20, 20).cuda())


@pytest.mark.parametrize("precision", ["exact16", "fast"])
@pytest.mark.parametrize("case", ["conv1_x40", "conv1_x0p002"])
def test_discriminator_parity_off_the_init_scale(precision, case, diag_dir):
    """VERDICT round 4, item 6a for the discriminator.  Its normalised layers are scale-free by construction (W / sigma), so the
    activation scale is set by the two layers without spectral norm: conv1 x 40 pushes every level of the U-Net to O(10 - 100) (hi
    tensors far outside [0, 1], scaled lo tensors of O(100), large gradients at the loss scale); conv1 x 0.002 pushes them to O(1e-3)
    (hi tensors at f16's subnormal border -- the pairs keep their precision, a single f16 does not).  Forward and every gradient
    against the float64 oracle, gates relative to the tensors' own scale."""
    d, sd, M = _make(precision, 11)
    f = 40.0 if case == "conv1_x40" else 0.002
    sd = {k: v.clone() for k, v in sd.items()}
    sd["conv1.weight"] = sd["conv1.weight"] * f
    sd["conv1.bias"] = sd["conv1.bias"] * f
    d.load_state_dict(sd)
    d.train()
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 48, 64, generator=gen)
    gw = torch.randn(2, 1, 48, 64, generator=gen)
    sdo = {k: v.double().clone() for k, v in sd.items()}
    for k in sdo:
        if not (k.endswith("_u") or k.endswith("_v")):
            sdo[k].requires_grad_(True)
    xo = x.double().clone().requires_grad_(True)
    yo = M.discriminator_forward(xo, sdo, True)
    (yo * gw.double()).sum().backward()
    xd = x.cuda().requires_grad_(True)
    y = d(xd)
    (y * gw.cuda()).sum().mul(256.0).backward()
    torch.cuda.synchronize()
    rel = lambda a, b: ((a.double() - b).norm() / b.norm().clamp_min(1e-300)).item()
    fwd = ((y.detach().cpu().double() - yo.detach()).abs().max() / yo.detach().abs().max()).item()
    errs = {name: rel(p.grad.cpu() / 256.0, sdo[name].grad) for name, p in d.named_parameters()}
    errs["x"] = rel(xd.grad.cpu() / 256.0, xo.grad)
    worst = max(errs, key=errs.get)
    vals = sorted(errs.values())
    rep = {"fwd_rel_max": fwd, "out_absmax": yo.detach().abs().max().item(), "worst": errs[worst], "worst_tensor": worst, "median": vals[len(vals) // 2]}
    with open(os.path.join(diag_dir, f"disc_offscale_{case}_{precision}.json"), "w") as fh:
        json.dump(rep, fh, indent=1)
    assert all(torch.isfinite(p.grad).all() for p in d.parameters()), rep
    if precision == "exact16":
        # a mask element may flip (pre-activations within rounding of zero): every tensor below it then moves by 1e-3 .. 1e-2
        assert fwd < 2e-5 and rep["median"] < 1e-4 and rep["worst"] < 2e-2, rep
    else:
        assert fwd < 2e-2 and rep["median"] < 5e-2 and rep["worst"] < 0.3, rep


def test_usm_sharp_backward_vs_oracle_autograd():
    from oracle import imgproc_ref as I
    from real_esrgan_pytorch_amd import imgproc
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 40, 56, generator=gen)
    gw = torch.randn(2, 3, 40, 56, generator=gen)
    xo = x.clone().requires_grad_(True)
    (I.usm_sharp(xo, I.usm_kernel(50, 0), 0.5, 10) * gw).sum().backward()
    usm = imgproc.USMSharp(50, 0).cuda()
    xd = x.cuda().requires_grad_(True)
    (usm(xd, 0.5, 10) * gw.cuda()).sum().backward()
    assert (xd.grad.cpu() - xo.grad).abs().max().item() < 2e-4 * xo.grad.abs().max().item()


def test_gan_step_vs_oracle():
    """One RealESRGAN step (train_realesrgan.py:459-521, perceptual term excluded) against the CPU oracle."""
    import torch.nn.functional as F
    import real_esrgan_pytorch_amd as R
    from oracle import imgproc_ref as I
    from oracle import model_ref as M
    from real_esrgan_pytorch_amd.train import RealESRGANStep
    nb = 1
    gsd = M.init_generator_state(31, bias_noise=0.02)
    gsd = {k: v for k, v in gsd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < nb}
    gsd["conv4.bias"] = gsd["conv4.bias"] + 0.5
    dsd = M.init_discriminator_state(32)
    gen = torch.Generator().manual_seed(3)
    lr = torch.rand(1, 3, 16, 16, generator=gen)
    hr = torch.rand(1, 3, 64, 64, generator=gen)
    # ---- oracle step
    gp = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    dp = {k: v.clone() for k, v in dsd.items()}
    for k in dp:
        if not (k.endswith("_u") or k.endswith("_v")):
            dp[k].requires_grad_(True)
    sr = M.generator_forward(lr, gp, 4, nb)
    pixel = F.l1_loss(I.usm_sharp(sr, I.usm_kernel(50, 0), 0.5, 10), hr)
    adv = 0.1 * F.binary_cross_entropy_with_logits(M.discriminator_forward(sr, dp, True), torch.ones(1, 1, 64, 64))
    (pixel + adv).backward()
    g_grads = {k: v.grad.clone() for k, v in gp.items()}
    for v in dp.values():
        v.grad = None
    d_hr = F.binary_cross_entropy_with_logits(M.discriminator_forward(hr, dp, True), torch.ones(1, 1, 64, 64))
    d_hr.backward()
    d_sr = F.binary_cross_entropy_with_logits(M.discriminator_forward(sr.detach().clone(), dp, True), torch.zeros(1, 1, 64, 64))
    d_sr.backward()
    # ---- MI355X step (lr = 0 so the weights stay put; Adam state/EMA still run)
    g = R.Generator(3, 3, 4, precision="strict", n_blocks=nb)
    g.load_state_dict(gsd)
    g = g.cuda()
    d = R.Discriminator(precision="strict")
    d.load_state_dict(dsd)
    d = d.cuda().train()
    ema = R.EMA(g, 0.999)
    ema.register()
    step = RealESRGANStep(g, d, ema, torch.optim.Adam(g.parameters(), 0.0, (0.9, 0.99)),
                          torch.optim.Adam(d.parameters(), 0.0, (0.9, 0.99)), scaler=None)
    out = step(hr.cuda(), lr.cuda())
    torch.cuda.synchronize()
    assert abs(out["pixel_loss"].item() - pixel.item()) < 1e-5
    assert abs(out["adversarial_loss"].item() - adv.item()) < 1e-5
    assert abs(out["d_loss_hr"].item() - d_hr.item()) < 1e-5
    assert abs(out["d_loss_sr"].item() - d_sr.item()) < 1e-5
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    for name, p in g.named_parameters():
        assert rel(p.grad.cpu(), g_grads[name]) < 2e-2, name
    for name, p in d.named_parameters():
        assert rel(p.grad.cpu(), dp[name].grad) < 2e-2, name
    assert torch.allclose(d.state_dict()["conv2.0.weight_u"].cpu(), dp["conv2.0.weight_u"], atol=1e-5)   # three power iterations


@pytest.mark.parametrize("precision,tol", [("strict", 1e-4), ("exact16", 1e-4), ("fast", 2e-2)])
def test_content_loss_forward_vs_oracle(precision, tol):
    from oracle import model_ref as M
    from real_esrgan_pytorch_amd.content_loss import ContentLoss
    nodes = ["features.2", "features.7", "features.16", "features.25", "features.34"]      # config.py:131
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    torch.manual_seed(4)
    for aliasing in (True, False):
        cl = ContentLoss(nodes, mean, std, precision=precision, inplace_relu_aliasing=aliasing).cuda()
        sd = {k: v.detach().cpu() for k, v in cl.state_dict().items() if k.startswith("features.")}
        gen = torch.Generator().manual_seed(6)
        sr, hr = torch.rand(2, 3, 64, 48, generator=gen), torch.rand(2, 3, 64, 48, generator=gen)
        ref = M.content_loss(sr, hr, sd, nodes, mean, std, aliasing)
        got = cl(sr.cuda(), hr.cuda())
        assert len(got) == 5
        for g, r in zip(got, ref):
            assert not g.requires_grad                              # detached, like torch.Tensor(...) at train_realesrgan.py:477
            assert abs(g.item() - r.item()) < tol * max(r.item(), 1e-6), (precision, aliasing, g.item(), r.item())


@pytest.mark.parametrize("precision", ["fast", "exact16"])
def test_vgg_grouped_launches_equal_one_launch_per_group(precision):
    """VGG19's 128..512-channel layers as ONE launch each (`ResrConvDesc.cout_groups` with a bias) against one launch per 64-channel
    group (RESR_VGG_PER_GROUP=1, read per call): the same tiles, weights and biases -- every tapped feature bit-equal, forward and
    through the native backward."""
    from real_esrgan_pytorch_amd.content_loss import _FeatureFn
    cl, sd, nodes, mean, std = _vgg_case(precision, True)
    gen = torch.Generator().manual_seed(21)
    x, other = torch.rand(2, 3, 64, 96, generator=gen).cuda(), torch.rand(2, 3, 64, 96, generator=gen).cuda()
    res = []
    for per_group in (False, True):
        if per_group:
            os.environ["RESR_VGG_PER_GROUP"] = "1"
        try:
            xd = x.clone().requires_grad_(True)
            outs = _FeatureFn.apply(cl, xd, other)
            sum(o.square().sum() for o in outs).mul(4.0).backward()
            torch.cuda.synchronize()
            res.append([o.detach().clone() for o in outs] + [xd.grad.clone()])
        finally:
            os.environ.pop("RESR_VGG_PER_GROUP", None)
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert res[0][-1].abs().sum().item() > 0


def _vgg_case(precision, aliasing):
    from real_esrgan_pytorch_amd.content_loss import ContentLoss
    nodes = ["features.2", "features.7", "features.16", "features.25", "features.34"]      # config.py:131
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    torch.manual_seed(4)
    cl = ContentLoss(nodes, mean, std, precision=precision, inplace_relu_aliasing=aliasing, detached=False).cuda()
    sd = {k: v.detach().cpu() for k, v in cl.state_dict().items() if k.startswith("features.")}
    return cl, sd, nodes, mean, std


@pytest.mark.parametrize("precision", ["strict", "exact16", "fast"])
@pytest.mark.parametrize("aliasing", [True, False])
def test_vgg_backward_vs_oracle_autograd(precision, aliasing, diag_dir):
    """The native VGG19 backward behind ContentLoss(detached=False) -- ReLU masks, 2x2 max-pool argmax, backward-data
    convolutions, no weight gradients (the VGG is frozen, model.py:306-308) -- against the oracle's autograd: random cotangents on
    the five tapped nodes, d/d(input) compared in relative L2; both settings of the inplace-ReLU aliasing of the taps.

    ReLU'(v) jumps at 0: a pre-activation within rounding of zero flips ONE mask element between two correct evaluations, and
    that one element costs ~1e-3 of the whole gradient at this size (tools/diag_content_loss.py shows the layer: every other
    pre-activation gradient agrees to 1e-6) -- the LeakyReLU-mask effect of test_gpu_generator.py.  Four inputs are run: every
    one must stay within 5e-3 (at most a couple of flips; measured 2.3-2.8e-3 with one); the flip-free ones show the arithmetic's own class, <= 2e-5 for strict
    and exact16 (measured 2e-6)."""
    from oracle import model_ref as M
    from real_esrgan_pytorch_amd.content_loss import _FeatureFn
    cl, sd, nodes, mean, std = _vgg_case(precision, aliasing)
    scale = 1.0 if precision == "strict" else 256.0                  # f16 (pair) gradients: a loss scale like GradScaler's
    rels = []
    for seed in (7, 8, 9, 10):
        gen = torch.Generator().manual_seed(seed)
        x, other = torch.rand(2, 3, 64, 48, generator=gen), torch.rand(2, 3, 64, 48, generator=gen)
        xo = x.clone().requires_grad_(True)
        fo = M.vgg_features(xo, sd, nodes, mean, std, aliasing)
        cot = {k: torch.randn(v.shape, generator=gen) / v[0].numel() ** 0.5 for k, v in fo.items()}       # NCHW
        sum((fo[k] * cot[k]).sum() for k in nodes).backward()
        xd = x.cuda().requires_grad_(True)
        outs = _FeatureFn.apply(cl, xd, other.cuda())               # NHWC fp32: the sr half (differentiable), then the hr half
        for i, k in enumerate(nodes):
            tol = 5e-2 if precision == "fast" else 1e-4
            assert (outs[i].detach().cpu().permute(0, 3, 1, 2) - fo[k].detach()).abs().max().item() < tol * max(1.0, fo[k].abs().max().item()), k
        (sum((outs[i] * cot[k].permute(0, 2, 3, 1).cuda()).sum() for i, k in enumerate(nodes)) * scale).backward()
        torch.cuda.synchronize()
        rels.append(((xd.grad.cpu() / scale - xo.grad).norm() / xo.grad.norm()).item())
    with open(os.path.join(diag_dir, f"vgg_backward_{precision}_{int(aliasing)}.json"), "w") as f:
        json.dump(rels, f)
    if precision == "fast":
        assert max(rels) < 0.15, rels                 # f16 activations through 16 layers: the fast mode's class
    else:
        assert max(rels) < 5e-3 and min(rels) < 2e-5, rels


@pytest.mark.parametrize("precision,tol", [("strict", 5e-3), ("exact16", 5e-3), ("fast", 0.15)])
@pytest.mark.parametrize("aliasing", [True, False])
def test_content_loss_gradient_vs_oracle_autograd(precision, tol, aliasing):
    """End to end: d(sum_i w_i L1(vgg_i(sr), vgg_i(hr))) / d(sr) of ContentLoss(detached=False) -- the perceptual term of the graph
    the reference wrote (model.py:311-335, weights config.py:137) -- against the oracle's autograd.  Besides the ReLU masks (see
    test_vgg_backward_vs_oracle_autograd) the L1 terms put a sign() on feature differences, which ties flip too (random VGG weights
    with zero biases leave many features at or next to 0 in both images): measured 2e-6 without a flip, 1-2e-3 with one."""
    from oracle import model_ref as M
    cl, sd, nodes, mean, std = _vgg_case(precision, aliasing)
    weights = [0.1, 0.1, 1.0, 1.0, 1.0]                                                   # config.py:137
    gen = torch.Generator().manual_seed(6)
    sr, hr = torch.rand(2, 3, 64, 48, generator=gen), torch.rand(2, 3, 64, 48, generator=gen)
    sro = sr.clone().requires_grad_(True)
    ref = M.content_loss(sro, hr, sd, nodes, mean, std, aliasing)
    sum(w * l for w, l in zip(weights, ref)).backward()
    scale = 1.0 if precision == "strict" else 4096.0
    srd = sr.cuda().requires_grad_(True)
    got = cl(srd, hr.cuda())
    for g, r in zip(got, ref):
        assert g.requires_grad
        assert abs(g.item() - r.item()) < (2e-2 if precision == "fast" else 1e-4) * max(r.item(), 1e-6)
    (sum(w * l for w, l in zip(weights, got)) * scale).backward()
    torch.cuda.synchronize()
    rel = ((srd.grad.cpu() / scale - sro.grad).norm() / sro.grad.norm()).item()
    assert rel < tol, (precision, aliasing, rel)
    # the default stays the reference's quirk: detached scalars, no gradient
    cl.detached = True
    assert not any(t.requires_grad for t in cl(sr.cuda().requires_grad_(True), hr.cuda()))


def test_exact16_backward_does_not_depend_on_the_loss_scale(monkeypatch):
    """The discriminator's native exact16 backward lifts a small incoming gradient by a power of two and unscales its results, like
    the generator's (csrc/disc_native.hip, common.h grad_prescale): under the BCE mean loss of the GAN step (g_y = (sigmoid - label)
    x scale / numel) weight gradients -- spectral-norm backward included -- and the input gradient at loss scales 1 and 2^16 are the
    same numbers times the scale bit for bit; without the lift the tiny-scale pass differs."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(6)
    d = R.Discriminator(precision="exact16").cuda().train()
    x = torch.rand(2, 3, 64, 64, device="cuda")

    def run(scale):
        d.zero_grad(set_to_none=True)
        xt = x.clone().requires_grad_(True)
        torch.manual_seed(1)      # (the power iteration draws nothing, but keep every call on the same footing)
        out = d(xt)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(out, torch.ones_like(out)) * scale
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in d.named_parameters() if p.grad is not None}, xt.grad.detach().clone()
    sd = {k: v.clone() for k, v in d.state_dict().items()}
    g1, x1 = run(1.0)
    d.load_state_dict(sd)         # the same u vectors for the second call
    g2, x2 = run(65536.0)
    assert len(g1) > 10
    assert all(torch.equal(g2[k], g1[k] * 65536.0) for k in g1)
    assert torch.equal(x2, x1 * 65536.0)
    monkeypatch.setenv("RESR_X2_NO_GRAD_PRESCALE", "1")
    d.load_state_dict(sd)
    g3, _ = run(1.0)
    monkeypatch.delenv("RESR_X2_NO_GRAD_PRESCALE")
    worst = max(((g3[k] - g1[k]).norm() / g1[k].norm().clamp_min(1e-30)).item() for k in g1)
    assert 0 < worst < 5e-2, worst      # pairs degrade gently (the lo halves carry 2^12 more range): visible, not catastrophic


def test_unknown_precision_raises():
    """An unsupported precision is an error, never a silent downgrade (round 2's exact16 -> fast)."""
    import real_esrgan_pytorch_amd as R
    with pytest.raises(ValueError):
        R.Discriminator(precision="bf16")
    assert R.Discriminator(precision="exact16").precision == "exact16"


def test_flat_parameter_mode_matches_per_tensor_mode():
    """`Discriminator.flat_parameter()`: one graph input / one gradient arena.  Two accumulated backward passes (the GAN step's
    D(hr) + D(sr), train_realesrgan.py:503-516), the frozen pass in between, zero_grad and a fused Adam step over the alias give
    the per-tensor mode's results bit for bit."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_discriminator_state(5)
    gen = torch.Generator().manual_seed(6)
    xa, xb = torch.rand(2, 3, 32, 40, generator=gen).cuda(), torch.rand(2, 3, 32, 40, generator=gen).cuda()

    def run(flat):
        d = R.Discriminator(precision="fast")
        d.load_state_dict(sd)
        d = d.cuda().train()
        params = [d.flat_parameter()] if flat else list(d.parameters())
        opt = torch.optim.Adam(params, 1e-3, (0.9, 0.99), fused=True)
        outs = []
        for _ in range(2):
            for p in list(d.parameters()) + ([d.flat_parameter()] if flat else []):       # the generator step's frozen pass
                p.requires_grad = False
            xg = xa.clone().requires_grad_(True)
            d(xg).sum().mul(64.0).backward()
            outs.append(xg.grad.clone())
            assert all(p.grad is None for p in d.parameters()) and (not flat or d.flat_parameter().grad is None)
            for p in list(d.parameters()) + ([d.flat_parameter()] if flat else []):
                p.requires_grad = True
            d.zero_grad(set_to_none=True)
            d(xa).sum().mul(64.0).backward()
            d(xb).square().sum().mul(64.0).backward()
            g = d.flat_grad().clone() if flat else torch.cat([p.grad.reshape(-1) for p in d.parameters()])
            outs.append(g)
            opt.step()
            d.zero_grad(set_to_none=True)
            assert not flat or d.flat_grad() is None
        outs.append(d.flat_parameters().detach().clone())
        outs.append(d.flat_uv().clone())
        return outs
    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


def test_exact16_weight_gradients_at_a_size_with_many_pixel_tiles():
    """exact16's default three tap-products per weight-gradient product at a size where the pixel splits are not capped by the
    tile count (>= 128 tiles of 8 x 32 pixels per level): the slab buffer of the workspace must hold jobs x 3 x splits slabs
    (round 3 sized the splits by the algorithmic product count: `discriminator: wgrad slabs` at 16 x 256^2).  Gradients against
    strict's (a flipped LeakyReLU mask element between two correct evaluations costs ~1e-2 in a tensor, see above)."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_discriminator_state(9)
    gen = torch.Generator().manual_seed(10)
    x = torch.rand(4, 3, 128, 128, generator=gen).cuda()
    gw = torch.randn(4, 1, 128, 128, generator=gen).cuda()
    grads = {}
    for precision, scale in (("strict", 1.0), ("exact16", 256.0)):
        d = R.Discriminator(precision=precision)
        d.load_state_dict(sd)
        d = d.cuda().train()
        (d(x) * gw).sum().mul(scale).backward()
        torch.cuda.synchronize()
        grads[precision] = {n: p.grad.clone() / scale for n, p in d.named_parameters()}
    for n, ref in grads["strict"].items():
        rel = ((grads["exact16"][n] - ref).norm() / ref.norm().clamp_min(1e-12)).item()
        assert rel < 2e-2, (n, rel)


def test_layer_mode_weight_gradients_equal_table_mode():
    """The 256..512-channel layers' weight gradients as ONE layer-mode launch pair each (wgrad.hip WgradLayer: the quad jobs'
    operands and slabs follow from their grid position, no job table) against the table-mode launch pairs
    (RESR_WGRAD_NO_LAYER_MODE=1, read per call): the same products in another split order -- equal to fp32 summation noise -- and
    against strict."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_discriminator_state(12)
    gen = torch.Generator().manual_seed(13)
    x = torch.rand(4, 3, 64, 96, generator=gen).cuda()
    gw = torch.randn(4, 1, 64, 96, generator=gen).cuda()

    def grads(precision, env):
        if env:
            os.environ["RESR_WGRAD_NO_LAYER_MODE"] = "1"
        try:
            d = R.Discriminator(precision=precision)
            d.load_state_dict(sd)
            d = d.cuda().train()
            scale = 1.0 if precision == "strict" else 256.0
            (d(x) * gw).sum().mul(scale).backward()
            torch.cuda.synchronize()
            return {n: p.grad.clone() / scale for n, p in d.named_parameters()}
        finally:
            os.environ.pop("RESR_WGRAD_NO_LAYER_MODE", None)
    layer, table, strict = grads("fast", False), grads("fast", True), grads("strict", False)
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    for n in layer:
        assert rel(layer[n], table[n]) < 2e-5, (n, rel(layer[n], table[n]))
        assert rel(layer[n], strict[n]) < 0.12, (n, rel(layer[n], strict[n]))
    # exact16: the same launch pairs on (hi, lo) operands, three tap-products per product
    layer16, table16 = grads("exact16", False), grads("exact16", True)
    for n in layer16:
        assert rel(layer16[n], table16[n]) < 5e-6, (n, rel(layer16[n], table16[n]))
        assert rel(layer16[n], strict[n]) < 5e-3, (n, rel(layer16[n], strict[n]))   # two fp32-class evaluations of a random-init network (a LeakyReLU flip or two apart): 1.5e-3 at worst, fast mode: 0.1
    assert sum(n.startswith(("down_block2", "down_block3", "up_block1")) for n in layer) == 3      # the three layers that take it


@pytest.mark.parametrize("shape", [(2, 7, 5, 64), (1, 1, 9, 8), (3, 6, 1, 16), (2, 16, 24, 128), (1, 2, 2, 8)])
def test_bilinear_up2x_quad_kernel_vs_torch_and_f32_kernel(shape):
    """resr_bilinear_up2x (F.interpolate(scale_factor=2, mode="bilinear", align_corners=False), discriminator_arch: model.py:170-184).
    The f16 forward computes four outputs per thread; the f32 forward one per thread.  Same fp32 arithmetic per output, so the f16
    result must be the ROUNDED f32 result bit for bit, on ragged shapes too (single rows / columns, odd sizes); both within fp32
    rounding of torch; the backward gather is the adjoint (torch autograd)."""
    import real_esrgan_pytorch_amd as R
    L = R._lib.lib()
    n, h, w, c = shape
    torch.manual_seed(h * 100 + w)
    src16 = torch.randn(n, h, w, c, device="cuda").half()
    src32 = src16.float()
    out16 = torch.empty(n, 2 * h, 2 * w, c, device="cuda", dtype=torch.half)
    out32 = torch.empty(n, 2 * h, 2 * w, c, device="cuda", dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    assert L.resr_bilinear_up2x(src16.data_ptr(), out16.data_ptr(), n, h, w, c, R._lib.RESR_F16, 0, st) == 0
    assert L.resr_bilinear_up2x(src32.data_ptr(), out32.data_ptr(), n, h, w, c, R._lib.RESR_F32, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(out16, out32.half())
    x = src32.permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    assert (out32 - ref.detach().permute(0, 2, 3, 1)).abs().max().item() <= 2e-6
    g32 = torch.randn(n, 2 * h, 2 * w, c, device="cuda")
    gin32 = torch.empty(n, h, w, c, device="cuda")
    assert L.resr_bilinear_up2x(g32.data_ptr(), gin32.data_ptr(), n, h, w, c, R._lib.RESR_F32, 1, st) == 0
    ref.backward(g32.permute(0, 3, 1, 2))
    torch.cuda.synchronize()
    assert (gin32 - x.grad.permute(0, 2, 3, 1)).abs().max().item() <= 1e-5
