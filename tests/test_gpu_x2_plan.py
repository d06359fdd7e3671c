"""-m gpu: exact16 with single-f16 growth planes / growth-plane gradients (ResrConvDesc.x2_pair_chunks, RESR_CONV_OUT_SINGLE,
ResrGeneratorDesc.x2_plan -- DESIGN.md section 2, round 5).

The residual stream of a dense block stays a (hi, lo) pair; the four growth planes (inference) and their gradients (backward) are
single f16 tensors: their chunks take two MFMA stages (x W0 + x W1) instead of three, conv1..conv4's weight gradients two
tap-products instead of three.  Reference arithmetic: /root/reference/model.py:87-98,255-272 (fp32 on the CPU, inference.py:52-53).
Later bits of the plan: the weight products read the growth planes as their hi tensor (bit 3), conv5's such products with g_y's hi tensor
(bit 4), and at inference the growth chunks meet the f16 weights alone -- one stage (bit 5, RESR_CONV_SINGLE_W16).  Default plan 59.
Gates: inference forward <= 2e-4 vs the fp32 oracle at 23 blocks for weights x 1, x 4 and after 60
training steps; every gradient tensor <= 1e-3 relative L2 vs the float64 evaluation of the oracle (shipped because the emulation
of tools/precision_ladder_sim.py keeps the worst tensor <= 5e-4 at three geometries and five seeds)."""
import ctypes as C
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
from tests.conftest import slow as _slow  # noqa: E402


@pytest.fixture(scope="module")
def U():
    from tests import gpu_util
    return gpu_util


def _pair_planar(t):
    """[N,C,H,W] fp32 (cpu) -> chunk-planar pair buffer [2][C/32][N,H,W,32] f16 (hi planes, then lo planes) + the pair's value."""
    n, c, h, w = t.shape
    hi = t.half()
    lo = ((t - hi.float()) * 4096.0).half()

    def planar(v):
        return v.reshape(n, c // 32, 32, h, w).permute(1, 0, 3, 4, 2).contiguous()
    buf = torch.stack([planar(hi), planar(lo)]).cuda()
    return buf, hi.double() + lo.double() / 4096.0, hi.double()


@pytest.mark.parametrize("name,cin,cout,n,h,w", [("growth_conv4", 160, 32, 3, 40, 36), ("closing_conv5", 192, 64, 2, 36, 70),
                                                 ("growth_conv2_small", 96, 32, 1, 12, 20)])
def test_conv_reads_single_chunks_and_writes_single_output(U, name, cin, cout, n, h, w, diag_dir):
    """One conv pass: chunks 0, 1 of the input are pairs, the chunks behind them single f16 tensors (their lo planes hold
    garbage that must not be read); a cout-32 pass stores a single f16 output (its lo plane stays untouched), the closing
    cout-64 pass stores a pair.  Against float64 on the values the kernel is given."""
    L = U.L
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    xb, xv, xhi = _pair_planar(x)
    xb[1, 2:] = 777.0                                    # lo planes of the single chunks: poison
    x_eff = torch.cat([xv[:, :64], xhi[:, 64:]], 1)
    plane = n * h * w * 32
    growth = cout == 32
    out = torch.full((2, cout // 32, n, h, w, 32), -7.0, dtype=torch.float16, device="cuda")
    d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16X2, 0, 1.0, 1.0, 1.0, 1.0, 0.2)
    d.in0_chunk_stride = plane
    d.out_chunk_stride = plane
    d.in0_lo_offset = (cin // 32) * plane
    d.out_lo_offset = (cout // 32) * plane
    d.x2_pair_chunks = 2
    ref = F.conv2d(x_eff, wt.double(), bias.double(), padding=1)
    res0 = None
    if growth:
        d.flags = L.CONV_LRELU | L.CONV_OUT_SINGLE
        ref = F.leaky_relu(ref, 0.2)
    else:
        r0 = torch.randn(n, cout, h, w, generator=g)
        res0, r0v, _ = _pair_planar(r0)
        d.res0_stride, d.res0_chunk_stride, d.s0, d.t0, d.res0_lo_offset = 32, plane, 0.2, 1.0, (cout // 32) * plane
        ref = ref * 0.2 + r0v
    packed = U.pack_conv(wt, L.RESR_F16X2)
    bias_d = bias.cuda()
    L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(xb), None, L.ptr(packed), L.ptr(bias_d), L.ptr(res0), None, None,
                                 L.ptr(out), None, L.stream_ptr()), "resr_conv3x3")
    torch.cuda.synchronize()

    def unplanar(t):
        return t.double().cpu().permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
    if growth:
        assert (out[1] == -7.0).all(), "a single f16 output must leave the lo tensor alone"
        got = unplanar(out[0])
        # the stored value is the f16 rounding of an fp32-class result: half an ulp of each element, not of the tensor's maximum
        bound = ref.abs() * (2.0 ** -11 * 1.01) + 2e-6
    else:
        got = unplanar(out[0]) + unplanar(out[1]) / 4096.0
        bound = torch.full_like(ref, 2e-6 * max(1.0, ref.abs().max().item()))
    err = (got - ref).abs()
    with open(os.path.join(diag_dir, f"x2_single_conv_{name}.json"), "w") as f:
        json.dump({"max_abs_err": err.max().item(), "ref_absmax": ref.abs().max().item(), "worst_over_bound": (err / bound).max().item()}, f)
    assert (err <= bound).all(), (name, err.max().item(), (err / bound).max().item())


def test_conv_mask_from_a_saved_activation_whose_hi_half_is_zero(U):
    """RESR_CONV_MASK without _BITS on RESR_F16X2 (the discriminator's backward-data passes): the mask is a saved activation, a
    (hi, lo) pair of out's shape.  Values below 2^-25 round their hi half to zero; their sign is then the lo half's (common.h
    pair_positive) -- a third of the mask elements here are +-1e-9.  Against float64."""
    L = U.L
    g = torch.Generator().manual_seed(3)
    n, cin, cout, h, w = 2, 64, 64, 20, 36
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    act = torch.randn(n, cout, h, w, generator=g)
    tiny = torch.rand(n, cout, h, w, generator=g) < 0.33
    act = torch.where(tiny, act.sign() * 1e-9, act)
    xb, xv, _ = _pair_planar(x)
    mb, mv, mhi = _pair_planar(act)
    assert (mhi[tiny] == 0).all() and (mb[1].cpu().float() != 0).any()
    plane = n * h * w * 32
    out = torch.full((2, cout // 32, n, h, w, 32), -7.0, dtype=torch.float16, device="cuda")
    d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 32, L.RESR_F16X2, L.CONV_MASK | L.CONV_NO_BIAS, 1.0, 1.0, 1.0, 1.0, 0.2)
    d.in0_chunk_stride = plane
    d.out_chunk_stride = plane
    d.mask_chunk_stride = plane
    d.in0_lo_offset = (cin // 32) * plane
    d.out_lo_offset = (cout // 32) * plane
    d.mask_lo_offset = (cout // 32) * plane
    packed = U.pack_conv(wt, L.RESR_F16X2)

    def run():
        out.fill_(-7.0)
        L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(xb), None, L.ptr(packed), None, None, None, L.ptr(mb), L.ptr(out), None,
                                     L.stream_ptr()), "resr_conv3x3")
        torch.cuda.synchronize()
    run()

    def unplanar(t):
        return t.double().cpu().permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
    got = unplanar(out[0]) + unplanar(out[1]) / 4096.0
    ref = F.conv2d(xv, wt.double(), None, padding=1) * torch.where(act.double() > 0, 1.0, 0.2)
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-6
    hi_only = F.conv2d(xv, wt.double(), None, padding=1) * torch.where(mhi > 0, 1.0, 0.2)      # what the hi halves alone give
    assert ((hi_only - ref).abs().max() / ref.abs().max()).item() > 0.1
    d.mask_lo_offset = 0          # the documented default: the mask's hi tensor alone (it may then be any f16 tensor)
    run()
    got0 = unplanar(out[0]) + unplanar(out[1]) / 4096.0
    assert ((got0 - hi_only).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.parametrize("cin,cout,n,h,w,splits", [(96, 32, 2, 40, 36, 4), (64, 64, 1, 24, 64, 2)])
def test_wgrad_single_g_equals_pair_g_with_zero_lo(U, cin, cout, n, h, w, splits):
    """g_lo_offset = 0: G is a single f16 tensor -- the (x_hi, g_lo) tap-product is not issued.  Same bits as the three-product
    launch on a G pair whose lo tensor is zero (the skipped slab only ever added zeros)."""
    L = U.L
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g)
    gy = torch.randn(n, cout, h, w, generator=g)

    def pair(t, zero_lo=False):
        hi = t.half()
        lo = ((t - hi.float()) * 4096.0).half()
        if zero_lo:
            lo = torch.zeros_like(lo)
        buf = torch.stack([hi.permute(0, 2, 3, 1), lo.permute(0, 2, 3, 1)]).contiguous().cuda()
        return buf, buf[0].numel()
    xb, x_lo = pair(x)
    gb, g_lo = pair(gy, zero_lo=True)

    def run(g_lo_offset, flags=0, expect=0):
        d = L.WgradDesc(n, h, w, cin, cin, cin, 0, cin, cout, cout, cout, L.RESR_F16X2, flags, splits, 1.0)
        d.x_lo_offset, d.g_lo_offset = x_lo, g_lo_offset
        partial = torch.zeros(L.lib().resr_wgrad_partial_bytes(C.byref(d)) // 4, device="cuda")
        dw = torch.full((cout, cin, 3, 3), -7.0, device="cuda")
        db = torch.full((cout,), -7.0, device="cuda")
        rc = L.lib().resr_conv3x3_wgrad(C.byref(d), L.ptr(xb), None, L.ptr(gb), L.ptr(partial), L.ptr(dw), L.ptr(db), L.stream_ptr())
        if expect != 0:
            assert rc == expect
            return None, None
        L.check(rc, "resr_conv3x3_wgrad")
        torch.cuda.synchronize()
        return dw, db
    dw3, db3 = run(g_lo)
    dw2, db2 = run(0, flags=L.CONV_OUT_SINGLE)       # the single-G mode is selected explicitly ...
    run(0, expect=-1)                               # ... a zero offset alone is a forgotten field
    assert torch.equal(dw3, dw2) and torch.equal(db3, db2)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    bs = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wt, bs, padding=1) * gy.half().double()).sum().backward()
    assert ((dw2.cpu().double() - wt.grad).norm() / wt.grad.norm()).item() < 5e-6


def _setup(n_blocks, seed, x2_plan, wscale=1.0, upscale=4):
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_generator_state(seed, 3, 3, upscale, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < n_blocks}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    if wscale != 1.0:   # off the init scale: the dense branches grow with their weights
        sd = {k: (v * wscale if k.endswith(".weight") and ".rdb" in k else v) for k, v in sd.items()}
    g = R.Generator(3, 3, upscale, precision="exact16", n_blocks=n_blocks, x2_plan=x2_plan)
    g.load_state_dict(sd)
    return g.cuda(), sd, M


@pytest.mark.parametrize("n,h,w,wscale", [(1, 24, 24, 1.0), pytest.param(8, 24, 24, 4.0, marks=_slow)])
def test_inference_plan_forward_vs_oracle(n, h, w, wscale, diag_dir):
    """23 blocks, eval: growth planes single f16 (50 stages per block) against the fp32 CPU oracle and against the all-pairs plan;
    weights at the reference's init scale and dense-block weights x 4 (activations grow, the dense branch is no longer small);
    8 x 24^2 runs the dense blocks as chained launches."""
    from real_esrgan_pytorch_amd import _lib as L
    g1, sd, M = _setup(23, 11, 1, wscale)
    g0, _, _ = _setup(23, 11, 0, wscale)
    g33, _, _ = _setup(23, 11, 33, wscale)      # + bit 5: the growth chunks take ONE stage (f16 weights): 40 stages per block
    x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y1 = g1.eval()(x.cuda()).cpu()
        y0 = g0.eval()(x.cuda()).cpu()
        y33 = g33.eval()(x.cuda()).cpu()
    yo = M.generator_forward(x, sd, 4, 23)
    yo64 = M.generator_forward(x.double(), {k: v.double() for k, v in sd.items()}, 4, 23)
    rep = {"plan1_vs_f32_oracle": (y1 - yo).abs().max().item(), "plan0_vs_f32_oracle": (y0 - yo).abs().max().item(),
           "plan1_vs_f64": (y1.double() - yo64).abs().max().item(), "plan0_vs_f64": (y0.double() - yo64).abs().max().item(),
           "plan33_vs_f32_oracle": (y33 - yo).abs().max().item(), "plan33_vs_f64": (y33.double() - yo64).abs().max().item(),
           "plan33_vs_plan1": (y33 - y1).abs().max().item(),
           "frac_unclamped": ((yo > 0) & (yo < 1)).float().mean().item()}
    with open(os.path.join(diag_dir, f"x2_plan_infer_{n}x{h}x{w}_w{wscale}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["plan1_vs_f32_oracle"] < 2e-4 and rep["plan1_vs_f64"] < 2e-4, rep
    assert rep["plan33_vs_f32_oracle"] < 2e-4 and rep["plan33_vs_f64"] < 2e-4 and rep["plan33_vs_plan1"] > 0, rep
    assert rep["plan0_vs_f64"] < 5e-5, rep
    assert int(L.lib().resr_debug_chain_errors()) == 0


@_slow
def test_inference_plan_after_training_steps(diag_dir):
    """The same gate on weights that have left the init: 60 RealESRNet steps (fast mode, as the train scripts run) on one fixed
    batch, then exact16 inference of the trained weights, plan 1 against plan 0 and the fp32 oracle."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import RealESRNetStep
    from oracle import model_ref as M
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast").cuda().train()
    opt = torch.optim.Adam(g.parameters(), 2e-4, (0.9, 0.99), fused=True)
    step = RealESRNetStep(g, None, opt, torch.amp.GradScaler("cuda"), None)
    gen = torch.Generator(device="cuda").manual_seed(1)
    hr = F.interpolate(torch.rand(4, 3, 32, 32, device="cuda", generator=gen), size=(256, 256), mode="bicubic").clamp(0, 1)
    lr = F.interpolate(hr, scale_factor=0.25, mode="area")
    losses = [step(hr, lr).item() for _ in range(60)]
    assert losses[-1] < 0.5 * losses[0]
    sd = {k: v.detach().float().cpu().clone() for k, v in g.state_dict().items()}
    ys = {}
    for plan in (0, 1, 33):
        ge = R.Generator(3, 3, 4, precision="exact16", x2_plan=plan)
        ge.load_state_dict(sd)
        with torch.no_grad():
            ys[plan] = ge.cuda().eval()(lr[:1]).cpu()
    yo = M.generator_forward(lr[:1].cpu(), sd, 4, 23)
    rep = {"plan1_vs_f32_oracle": (ys[1] - yo).abs().max().item(), "plan0_vs_f32_oracle": (ys[0] - yo).abs().max().item(),
           "plan1_vs_plan0": (ys[1] - ys[0]).abs().max().item(), "plan33_vs_f32_oracle": (ys[33] - yo).abs().max().item()}
    with open(os.path.join(diag_dir, "x2_plan_infer_trained.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["plan1_vs_f32_oracle"] < 2e-4 and rep["plan33_vs_f32_oracle"] < 2e-4, rep


@pytest.mark.parametrize("n,h,w,n_blocks,seed", [pytest.param(1, 24, 24, 23, 11, marks=_slow), (8, 32, 32, 3, 11), (2, 33, 17, 2, 13)])   # (seeds 12, 13 at full depth: tools/x2_plan_validate.py)
def test_training_plan_gradients_vs_float64_oracle(n, h, w, n_blocks, seed, diag_dir):
    """Backward with single-f16 growth-plane gradients (x2_plan bit 1) and weight products that read the growth planes as their hi
    tensor (bits 3, 4) -- the default plan, 27; the forward pass keeps every pair: all gradient tensors
    against the float64 evaluation of the oracle AND against the all-pairs plan on the same device.  Gate 1e-3 relative L2 per
    tensor (emulation: worst 3-5e-4); the forward pass and the input gradient stay at the all-pairs level.
    The two plans share their forward pass bit for bit, hence their LeakyReLU masks: the distance between them is the rung's own
    effect.  Against float64 a pre-activation within rounding of zero may flip a mask element in ANY finite-precision forward (the
    fp32 CPU path does it on seed 11; exact16 does it on seed 12 -- one element of trunk.6.rdb1.conv3, the
    very element the emulation's rounded-growth-plane forward flips: 1.011e-2 in both): such a tensor is accepted when the
    all-pairs plan shows the same distance."""
    g, sd, M = _setup(n_blocks, seed, 3 if (n, seed) == (8, 11) else 27)     # the default plan (one case: without bits 3, 4)
    g0, _, _ = _setup(n_blocks, seed, 0)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen)
    gw = torch.randn(n, 3, 4 * h, 4 * w, generator=gen)
    sdo = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.double().clone().requires_grad_(True)
    yo = M.generator_forward(xo, sdo, 4, n_blocks)
    (yo * gw.double()).sum().backward()

    def run(model):
        xd = x.cuda().requires_grad_(True)
        y = model.train()(xd)
        (y * gw.cuda()).sum().mul(1024.0).backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), {name: p.grad.cpu().double() / 1024.0 for name, p in model.named_parameters()}, xd.grad.cpu().double() / 1024.0
    y, gr, gx = run(g)
    y0, gr0, gx0 = run(g0)
    assert torch.equal(y, y0), "the training forward does not depend on the plan"

    def rel(got, ref):
        return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    errs = {name: rel(gr[name], sdo[name].grad) for name in gr}
    errs0 = {name: rel(gr0[name], sdo[name].grad) for name in gr}
    between = {name: rel(gr[name], gr0[name]) for name in gr}
    worst, worst_b = max(errs, key=errs.get), max(between, key=between.get)
    vals = sorted(errs.values())
    rep = {"fwd_vs_f64": (y.double() - yo.detach()).abs().max().item(), "worst_vs_f64": errs[worst], "worst_tensor": worst,
           "all_pairs_plan_on_that_tensor": errs0[worst], "worst_all_pairs_vs_f64": max(errs0.values()),
           "worst_vs_all_pairs_plan": between[worst_b], "worst_vs_all_pairs_tensor": worst_b,
           "median_vs_f64": vals[len(vals) // 2], "gx_vs_f64": rel(gx, xo.grad), "gx_vs_all_pairs": rel(gx, gx0)}
    with open(os.path.join(diag_dir, f"x2_plan_train_{n}x{h}x{w}_{n_blocks}_{seed}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["fwd_vs_f64"] < 5e-5, rep
    assert rep["worst_vs_all_pairs_plan"] < 1e-3 and rep["gx_vs_all_pairs"] < 1e-4, rep
    for name, e in errs.items():     # vs float64: inside the gate, or a mask flip of the (shared) forward pass
        assert e < 1e-3 or errs0[name] > 0.8 * e, (name, e, errs0[name])
    assert rep["worst_vs_all_pairs_plan"] > 5e-5, "the plan does not seem to be active (gradients at the all-pairs level)"
    # the bias gradients sum hi + lo of the growth-plane gradients: they stay at the level of the propagated error alone
    worst_bias = max(e for name, e in between.items() if name.endswith(".bias"))
    rep["worst_bias_vs_all_pairs"] = worst_bias
    assert worst_bias < 2e-4, rep


def test_plan_bits_are_honoured_and_ignored_outside_exact16():
    """x2_plan = 0 reproduces the all-pairs results bit for bit whatever $RESR_X2_PLAN says; a training forward ignores bit 0;
    fast mode ignores the plan."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(2)
    ref = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=0).cuda()
    sd = ref.state_dict()
    x = torch.rand(8, 3, 24, 32, device="cuda")
    outs = {}
    for plan in (0, 1, 2, 3):
        g = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=plan).cuda()
        g.load_state_dict(sd)
        with torch.no_grad():
            ye = g.eval()(x)
        xt = x.clone().requires_grad_(True)
        yt = g.train()(xt)
        yt.square().sum().mul(256.0).backward()
        outs[plan] = (ye, yt.detach(), xt.grad.clone(), g.trunk[0].rdb1.conv2.weight.grad.clone())
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][1], outs[3][1]), "a training forward keeps every pair"
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[1][0], outs[3][0])
    assert not torch.equal(outs[0][0], outs[1][0]), "bit 0 changes the inference forward"
    assert torch.equal(outs[0][3], outs[1][3]) and not torch.equal(outs[0][3], outs[2][3]), "bit 1 changes the backward pass"
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-4
    f0 = R.Generator(3, 3, 4, precision="fast", n_blocks=2, x2_plan=0).cuda()
    f3 = R.Generator(3, 3, 4, precision="fast", n_blocks=2, x2_plan=3).cuda()
    f0.load_state_dict(sd)
    f3.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(f0(x), f3(x))
    with pytest.raises(ValueError):
        R.Generator(3, 3, 4, precision="exact16", x2_plan=64)
    # bit 3: the weight products read the growth planes (X chunks 2..) as their hi tensor -- conv2's gradient changes, conv1's
    # (stream chunks only) and the backward-data path (the input gradient) do not
    g11 = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=11).cuda()
    g11.load_state_dict(sd)
    xt = x.clone().requires_grad_(True)
    y11 = g11.train()(xt)
    y11.square().sum().mul(256.0).backward()
    assert torch.equal(y11.detach(), outs[3][1]) and torch.equal(xt.grad, outs[3][2])
    g3 = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=3).cuda()
    g3.load_state_dict(sd)
    xt3 = x.clone().requires_grad_(True)
    g3.train()(xt3).square().sum().mul(256.0).backward()
    same = lambda a, b: ((a - b).norm() / b.norm()).item() < 2e-6      # (the pixel splits of a launch follow its job count: other partial sums)
    assert same(g11.trunk[0].rdb1.conv1.weight.grad, g3.trunk[0].rdb1.conv1.weight.grad)
    assert same(g11.trunk[0].rdb1.conv2.bias.grad, g3.trunk[0].rdb1.conv2.bias.grad)
    for name in ("conv2", "conv5"):
        a, b = getattr(g11.trunk[0].rdb1, name).weight.grad, getattr(g3.trunk[0].rdb1, name).weight.grad
        assert not torch.equal(a, b) and ((a - b).norm() / b.norm()).item() < 2e-4, name     # (3e-7 here: the growth planes are the small operand)
    # bit 4 (with bit 3): conv5's growth-plane products take g_y's hi tensor alone -- conv5's gradient moves again, conv2's does not
    g27 = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=27).cuda()
    g27.load_state_dict(sd)
    xt = x.clone().requires_grad_(True)
    g27.train()(xt).square().sum().mul(256.0).backward()
    assert torch.equal(xt.grad, outs[3][2])
    assert same(g27.trunk[0].rdb1.conv2.weight.grad, g11.trunk[0].rdb1.conv2.weight.grad)
    a, b = g27.trunk[0].rdb1.conv5.weight.grad, g11.trunk[0].rdb1.conv5.weight.grad
    assert not torch.equal(a, b) and ((a - b).norm() / b.norm()).item() < 2e-4
    assert same(g27.trunk[0].rdb1.conv5.bias.grad, g11.trunk[0].rdb1.conv5.bias.grad)
    # bit 2 (opt-in): the growth-plane gradients stored single as well -- another backward pass, the same forward
    g7 = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=7).cuda()
    g7.load_state_dict(sd)
    xt = x.clone().requires_grad_(True)
    y7 = g7.train()(xt)
    y7.square().sum().mul(256.0).backward()
    w7 = g7.trunk[0].rdb1.conv2.weight.grad
    assert torch.equal(y7.detach(), outs[3][1]) and not torch.equal(w7, outs[3][3])
    assert not torch.equal(g7.trunk[0].rdb1.conv2.bias.grad, g3.trunk[0].rdb1.conv2.bias.grad)
    assert ((w7 - outs[0][3]).norm() / outs[0][3].norm()).item() < 1e-3


def _l1_grads(g, plan, x, target, scale, poison=None):
    g.x2_plan = plan
    g.zero_grad(set_to_none=True)
    xt = x.clone().requires_grad_(True)
    y = g(xt)
    loss = (y - target).abs().mean() * scale
    if poison is None:
        loss.backward()
    else:
        gy = torch.autograd.grad(loss, y)[0]
        gy.view(-1)[poison[0]] = poison[1]
        y.backward(gy)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in g.named_parameters()}, xt.grad.detach().clone()


def test_exact16_backward_does_not_depend_on_the_loss_scale(diag_dir, monkeypatch):
    """exact16 runs its backward pass on g_y * 2^k (max |.| in [1, 2)) and unscales every result (generator.hip, common.h
    grad_prescale): under the train step's L1 mean loss (g_y = scale / numel per element: 4e-6 here at scale 1, 2e-8 at the headline
    geometry) the gradients at loss scales 1, 2^10 and 2^20 are the same numbers times the scale, BIT FOR BIT, and the default plan
    (growth-plane gradients read as their f16 hi halves) stays at its usual distance from the all-pairs plan.  Without the
    pre-scale those hi halves are f16 subnormals at small scales: the same comparison is off by percents
    (tools/x2_plan_validate.py: 0.82 at 16 x 256^2 and loss scale 2^10, 3.3e-3 at a GradScaler's initial 2^16)."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(4)
    g = R.Generator(3, 3, 4, precision="exact16", n_blocks=23).cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    gen = torch.Generator(device="cuda").manual_seed(9)
    n, s = 4, 32
    x = torch.rand(n, 3, s, s, device="cuda", generator=gen)
    target = torch.rand(n, 3, 4 * s, 4 * s, device="cuda", generator=gen)
    base, gx_base = _l1_grads(g, 27, x, target, 1.0)
    for scale in (2.0 ** 10, 2.0 ** 20):
        gr, gx = _l1_grads(g, 27, x, target, scale)
        assert all(torch.equal(gr[k], base[k] * scale) for k in base), f"loss scale {scale}"
        assert torch.equal(gx, gx_base * scale)
    ref, gx_ref = _l1_grads(g, 0, x, target, 1.0)

    def worst(a, b):
        return max(((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-300)).item() for k in b)
    rep = {"plan3_vs_plan0_loss_scale_1": worst(base, ref), "gx": ((gx_base - gx_ref).norm() / gx_ref.norm()).item()}
    monkeypatch.setenv("RESR_X2_NO_GRAD_PRESCALE", "1")
    raw, _ = _l1_grads(g, 27, x, target, 1.0)
    raw0, _ = _l1_grads(g, 0, x, target, 1.0)
    monkeypatch.delenv("RESR_X2_NO_GRAD_PRESCALE")
    rep["without_prescale_plan3_vs_plan0"] = worst(raw, raw0)
    rep["without_prescale_plan0_vs_prescaled_plan0"] = worst(raw0, ref)
    with open(os.path.join(diag_dir, "x2_prescale.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["plan3_vs_plan0_loss_scale_1"] < 2e-4 and rep["gx"] < 1e-5, rep
    assert rep["without_prescale_plan3_vs_plan0"] > 1e-2, ("the case does not exercise f16 subnormals", rep)


def test_fast_backward_does_not_depend_on_the_loss_scale_either():
    """The lift applies to both 16-bit modes: fast mode's gradients under the L1 mean loss at loss scales 1 and 2^20 are the same
    numbers times the scale bit for bit (without it the scale-1 pass is f16 underflow: tools/fast_loss_scale_probe.py)."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(4)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=4).cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    gen = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(8, 3, 32, 32, device="cuda", generator=gen)
    target = torch.rand(8, 3, 128, 128, device="cuda", generator=gen)
    base, gx_base = _l1_grads(g, 0, x, target, 1.0)
    gr, gx = _l1_grads(g, 0, x, target, 2.0 ** 20)
    assert all(torch.equal(gr[k], base[k] * 2.0 ** 20) for k in base) and torch.equal(gx, gx_base * 2.0 ** 20)
    assert all(v.abs().max() > 0 for v in base.values())


def test_exact16_backward_keeps_a_non_finite_gradient_visible():
    """The pre-scale takes its factor from the bits of max |g_y|: an inf or NaN element wins the maximum, the factor falls back to 1
    and the non-finite value reaches the weight gradients -- a GradScaler's inf check still skips the step."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(4)
    g = R.Generator(3, 3, 4, precision="exact16", n_blocks=2).cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)      # keep the outputs inside the clamp: its backward mask would drop the poisoned element otherwise
    x = torch.rand(2, 3, 24, 24, device="cuda")
    target = torch.rand(2, 3, 96, 96, device="cuda")
    clean, _ = _l1_grads(g, 27, x, target, 1024.0)
    assert all(torch.isfinite(v).all() for v in clean.values())
    for bad in (float("inf"), float("nan")):
        gr, _ = _l1_grads(g, 27, x, target, 1024.0, poison=(5000, bad))
        assert not torch.isfinite(gr["conv4.weight"]).all(), bad
    again, _ = _l1_grads(g, 27, x, target, 1024.0)
    assert all(torch.equal(again[k], clean[k]) for k in clean), "the next pass is clean again"


@pytest.mark.parametrize("precision", ["fast", "exact16"])
def test_lifted_backward_backs_off_when_the_gradient_outgrows_its_headroom(precision):
    """the lift re-raises max |g_y| to [2^6, 2^7) on every step, so a gradient that grows by more than f16's
    remaining 2^9 on its way back would overflow at EVERY loss scale -- a GradScaler halving its scale could not cure it.  Here the
    backward gain of the HR tail is 2^16 by construction (conv4's and conv3's weights x 2^8 each -- below exact16's |w| < 16 --,
    upsampling2's weights and bias x 2^-16; fast mode, whose f16 activations could not hold the 2^-16, takes 2^11 on conv4 against conv3: LeakyReLU is positively homogeneous, so the forward pass is the same function): the first
    lifted passes overflow, set the flag in the workspace's pre-scale slot, and every later pass aims 2^4 lower (common.h) -- after at
    most five skipped steps the GradScaler's scale stops decaying and the weights move.  Without the back-off every step of the loop
    below is skipped."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(2)
    g = R.Generator(3, 3, 4, precision=precision, n_blocks=1).cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
        if precision == "fast":
            g.conv4.weight.mul_(2.0 ** 11)
            g.conv3[0].weight.mul_(2.0 ** -11)
            g.conv3[0].bias.mul_(2.0 ** -11)
        else:
            g.conv4.weight.mul_(2.0 ** 8)
            g.conv3[0].weight.mul_(2.0 ** 8)
            g.conv3[0].bias.mul_(2.0 ** -8)
            g.upsampling2[0].weight.mul_(2.0 ** -16)
            g.upsampling2[0].bias.mul_(2.0 ** -16)
            assert g.conv4.weight.abs().max() < 16 and g.conv3[0].weight.abs().max() < 16
    opt = torch.optim.Adam(g.parameters(), 1e-6, (0.9, 0.99))
    scaler = torch.amp.GradScaler("cuda")
    gen = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(4, 3, 32, 32, device="cuda", generator=gen)
    target = torch.rand(4, 3, 128, 128, device="cuda", generator=gen)
    w0 = g.conv1.weight.detach().clone()
    scales, finite = [], []
    for _ in range(10):
        opt.zero_grad(set_to_none=True)
        scaler.scale((g(x) - target).abs().mean()).backward()
        finite.append(all(torch.isfinite(p.grad).all().item() for p in g.parameters()))
        scaler.step(opt)
        scaler.update()
        scales.append(scaler.get_scale())
    assert not finite[0], "the case does not overflow the lifted pass: no back-off exercised"
    assert all(finite[5:]), (finite, scales)
    assert scales[-1] >= 65536.0 / 32, (finite, scales)     # at most five skipped steps; without the back-off: 65536 / 2^10
    assert not torch.equal(g.conv1.weight.detach(), w0), "no optimizer step was taken"


@pytest.mark.parametrize("plan", [3, 35])
@pytest.mark.parametrize("n,h,w", [(8, 24, 40), (16, 64, 64), (16, 128, 128)])
def test_single_plane_chains_equal_separate_launches(n, h, w, plan):
    """Chained dense-block launches with two-stage chunks (the dependent chunk of a job is its last TWO stages): inference forward
    (x2_plan bit 0, conv5 inside the chain on the small launches) and training backward (bit 1) bit-equal to one launch per pass."""
    import real_esrgan_pytorch_amd as R
    L = R._lib
    torch.manual_seed(3)
    g = R.Generator(3, 3, 4, precision="exact16", n_blocks=1, x2_plan=plan).cuda()     # 35: the growth chunks of the inference forward take ONE stage
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(n, 3, h, w, device="cuda", generator=gen)
    gw = torch.randn(n, 3, 4 * h, 4 * w, device="cuda", generator=gen)

    def run(no_chain):
        if no_chain:
            os.environ["RESR_CONV_NO_CHAIN"] = "1"
        try:
            with torch.no_grad():
                ye = g.eval()(x).clone()
            g.zero_grad(set_to_none=True)
            xt = x.clone().requires_grad_(True)
            (g.train()(xt) * gw).sum().mul(256.0).backward()
            torch.cuda.synchronize()
            return ye, xt.grad.clone(), [p.grad.clone() for p in g.parameters()]
        finally:
            os.environ.pop("RESR_CONV_NO_CHAIN", None)
    y0, gx0, g0 = run(True)
    for rep in range(2):
        y1, gx1, g1 = run(False)
        assert torch.equal(y0, y1), (rep, (y0 - y1).abs().max().item())
        assert torch.equal(gx0, gx1), rep
        for i, (a, b) in enumerate(zip(g0, g1)):
            assert torch.equal(a, b), (rep, i, (a - b).abs().max().item())
    assert int(L.lib().resr_debug_chain_errors()) == 0


# ---- off the init scale: every parity case above uses 0.1 x kaiming dense blocks and [0, 1] images -----------
def _oracle_grads(M, sd, x, gw, n_blocks, dt=torch.float64):
    sdo = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.to(dt).clone().requires_grad_(True)
    yo = M.generator_forward(xo, sdo, 4, n_blocks)
    (yo * gw.to(dt)).sum().backward()
    return yo.detach(), {k: v.grad for k, v in sdo.items()}, xo.grad


_OFF_INIT_ORACLE = {}


@pytest.mark.parametrize("precision,plan", [("exact16", 27 + 128 + 512), pytest.param("exact16", 0, marks=_slow), ("fast", 0)])
@pytest.mark.parametrize("case", ["dense_x4", "stream_x40", "stream_x0p01"])
def test_generator_parity_off_the_init_scale(precision, plan, case, diag_dir):
    """Forward + backward of a 6-block generator against the float64 oracle with (a) the dense-block weights x 4 (the branches are
    no longer a small correction of the stream) and (b) conv1's weights x 40 (stream and growth planes of O(10 - 100): the hi
    tensors leave [0, 1], the scaled lo tensors reach 2^12 x 2^-11 x 100 ~ 200, the loss scale meets large gradients).  exact16
    (default plan and all-pairs) keeps its gates RELATIVE to the activation scale; fast keeps f16's class."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    nb = 6
    sd = M.init_generator_state(17, 3, 3, 4, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < nb}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    if case == "dense_x4":
        sd = {k: (v * 4.0 if k.endswith(".weight") and ".rdb" in k else v) for k, v in sd.items()}
    elif case == "stream_x40":
        sd["conv1.weight"] = sd["conv1.weight"] * 40.0
        sd["conv4.weight"] = sd["conv4.weight"] / 40.0          # keeps the output inside the clamp, the gradients large
    else:
        # a stream of O(1e-3): hi tensors near / inside f16's subnormal range (6e-5), lo tensors (scaled by 2^12) still normal --
        # the pair keeps ~1e-11 absolute whatever the magnitude; a single f16 (fast) does not
        for k in ("conv1.weight", "conv1.bias"):
            sd[k] = sd[k] * 0.002
    g = R.Generator(3, 3, 4, precision=precision, n_blocks=nb, x2_plan=plan)
    g.load_state_dict(sd)
    g = g.cuda().train()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 24, 28, generator=gen)
    gw = torch.randn(2, 3, 96, 112, generator=gen)
    if case not in _OFF_INIT_ORACLE:      # (one float64 evaluation per case: the three precision / plan rows share it)
        _OFF_INIT_ORACLE[case] = _oracle_grads(M, sd, x, gw, nb)
    yo, go, gxo = _OFF_INIT_ORACLE[case]
    with torch.no_grad():      # the stream's magnitude, for the record
        f = torch.nn.functional.conv2d(x.double(), sd["conv1.weight"].double(), sd["conv1.bias"].double(), padding=1)
    scale = 256.0
    xd = x.cuda().requires_grad_(True)
    y = g(xd)
    (y * gw.cuda()).sum().mul(scale).backward()
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double() - b).norm() / b.norm().clamp_min(1e-30)).item()
    errs = {k: rel(p.grad.cpu() / scale, go[k]) for k, p in g.named_parameters()}
    worst = max(errs, key=errs.get)
    rep = {"stream_absmax": f.abs().max().item(), "fwd": (y.detach().cpu().double() - yo).abs().max().item(), "worst": errs[worst], "worst_tensor": worst,
           "median": sorted(errs.values())[len(errs) // 2], "gx": rel(xd.grad.cpu() / scale, gxo), "unclamped": ((yo > 0) & (yo < 1)).double().mean().item()}
    with open(os.path.join(diag_dir, f"offscale_{case}_{precision}_{plan}.json"), "w") as fjson:
        json.dump(rep, fjson, indent=1)
    assert torch.isfinite(y).all() and all(torch.isfinite(p.grad).all() for p in g.parameters())
    if case == "stream_x0p01":
        # A stream of O(1e-3) costs the pair format nothing (f16 subnormals are multiplied exactly:
        # test_exact16_subnormal_operands_are_exact).  What this case shows instead, in BOTH exact16 plans and whatever the loss
        # scale (tools/diag_smallscale.py): one LeakyReLU mask element of `upsampling1` flips against float64 -- every tensor BELOW
        # that layer sits at 1.3-3e-3, upsampling2 / conv3 / conv4 at 1e-6, the forward pass at 1e-6.  With 1344 LR pixels one element
        # is 1e-3 of a tensor; the flip-aware gates of test_training_plan_gradients_vs_float64_oracle apply: here the tensors above
        # every trunk LeakyReLU stay at rounding level and nothing exceeds the signature of a single flip.
        assert rep["fwd"] < 2e-4, rep
        if precision == "exact16":
            assert max(errs[k] for k in errs if k.startswith(("conv3", "conv4"))) < 1e-5, rep
            assert rep["worst"] < 1e-2, rep
        return
    if precision == "exact16":
        assert rep["fwd"] < 2e-4, rep
        # (a mask flip of the forward pass would show as ~1e-2 on one tensor in BOTH plans: see test_training_plan_gradients_vs_float64_oracle)
        assert rep["worst"] < (1e-3 if plan else 5e-5) or rep["worst"] > 3e-3, rep
        assert rep["median"] < (5e-4 if plan else 2e-5) and rep["gx"] < 1e-4, rep
    else:
        assert rep["fwd"] < 2e-2 and rep["median"] < 0.15, rep


def test_exact16_weight_overflow_is_loud():
    """exact16 packs W0 = f16(w * 2^12): |w| >= 16 overflows (include/resr.h).  That must never be silent: W0 = inf and its remainder
    W1 = -inf meet in the accumulator as NaN, so the output -- and a training loss -- turns NaN; |w| just below the limit is exact."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(1)
    g = R.Generator(3, 3, 4, precision="exact16", n_blocks=1).cuda().eval()
    x = torch.rand(1, 3, 16, 16, device="cuda")
    with torch.no_grad():
        y0 = g(x)
        g.trunk[0].rdb1.conv1.weight[0, 0, 1, 1] = 15.9
        y1 = g(x)
        g.trunk[0].rdb1.conv1.weight[0, 0, 1, 1] = 17.0
        y2 = g(x)
    assert torch.isfinite(y0).all() and torch.isfinite(y1).all()
    assert torch.isnan(y2).any(), "a weight beyond the split format's range must poison the output"


def test_exact16_subnormal_operands_are_exact(U, diag_dir):
    """A question the small-scale case raised, answered by measurement: does `v_mfma_f32_32x32x16_f16` keep f16 SUBNORMAL operands?
    One 64 -> 32 convolution on pair operands whose values sit (a) in f16's normal range, (b) 99.8 % below 6.1e-5 (hi tensors
    subnormal; lo tensors, scaled by 2^12, normal).  It does: (b) is as exact as (a) -- relative L2 3.9e-7 against float64 (3.1e-7
    for (a)) -- so the pair format's accuracy is relative down to f16's smallest subnormal, not absolute."""
    L = U.L
    g = torch.Generator().manual_seed(9)
    n, cin, cout, h, w = 1, 64, 32, 24, 32
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    rep = {}
    for tag, scale in (("normal", 1.0), ("tiny", 2e-5)):
        x = torch.randn(n, cin, h, w, generator=g) * scale
        xb, xv, _ = _pair_planar(x)
        plane = n * h * w * 32
        out = torch.zeros((2, 1, n, h, w, 32), dtype=torch.float16, device="cuda")
        d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16X2, L.CONV_NO_BIAS, 1.0, 1.0, 1.0, 1.0, 0.2)
        d.in0_chunk_stride, d.out_chunk_stride = plane, plane
        d.in0_lo_offset, d.out_lo_offset = 2 * plane, plane
        L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(xb), None, L.ptr(U.pack_conv(wt, L.RESR_F16X2)), None, None, None, None, L.ptr(out), None,
                                     L.stream_ptr()), "resr_conv3x3")
        torch.cuda.synchronize()
        got = (out[0].double() + out[1].double() / 4096.0).cpu().permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
        ref = F.conv2d(xv, wt.double(), padding=1)
        rep[tag] = {"rel_l2": ((got - ref).norm() / ref.norm()).item(), "max_abs": (got - ref).abs().max().item(), "ref_absmax": ref.abs().max().item(),
                    "share_of_hi_below_6e-5": (xv.abs() < 6.1e-5).double().mean().item()}
    with open(os.path.join(diag_dir, "x2_subnormal_probe.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["normal"]["rel_l2"] < 2e-6 and rep["tiny"]["rel_l2"] < 2e-6, rep
