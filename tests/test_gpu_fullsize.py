"""Size-independent properties at BASELINE.json's full geometry (LR 256^2 -> HR 1024^2, 23 RRDBs, fast mode), where the
CPU oracle would take minutes: batch independence (bit-exact), run-to-run determinism (bit-exact, forward and all 702
gradients), linearity of the backward pass in the loss scale, and additivity of the weight gradients over the batch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen():
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(3)
    g = R.Generator(3, 3, 4, precision="fast").cuda()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)      # centre the output inside the clamp range so gradients flow
    return g


def _grads(g, x, gy, scale):
    g.zero_grad(set_to_none=True)
    y = g(x)
    (y * gy).sum().mul(scale).backward()
    torch.cuda.synchronize()
    return y.detach().clone(), g.flat_grad().detach().clone()


def test_full_geometry_properties(gen):
    g = gen
    rng = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(2, 3, 256, 256, device="cuda", generator=rng)
    gy = torch.randn(2, 3, 1024, 1024, device="cuda", generator=rng) / 1024.0
    g.train()
    y_a, gr_a = _grads(g, x, gy, 1024.0)
    y_b, gr_b = _grads(g, x, gy, 1024.0)
    assert y_a.shape == (2, 3, 1024, 1024) and torch.isfinite(y_a).all() and torch.isfinite(gr_a).all()
    assert 0.05 < ((y_a > 0) & (y_a < 1)).float().mean().item()
    # determinism: same tiles, same accumulation order, fixed-order slab reduction
    assert torch.equal(y_a, y_b) and torch.equal(gr_a, gr_b)
    # batch independence: every output pixel's arithmetic does not depend on which other images share the launch
    y0, gr0 = _grads(g, x[:1], gy[:1], 1024.0)
    y1, gr1 = _grads(g, x[1:], gy[1:], 1024.0)
    assert torch.equal(y0, y_a[:1]) and torch.equal(y1, y_a[1:])
    # weight gradients are sums over the batch: fp32 slab sums in a different split order -> tight relative tolerance
    rel = ((gr0 + gr1) - gr_a).norm() / gr_a.norm()
    assert rel.item() < 2e-3, rel.item()     # f16 activation-gradients, fp32 accumulation: ~1e-4 observed
    # backward is linear in the incoming gradient; with f16 activation-gradients a power-of-two loss scale only moves
    # which values fall into the f16 subnormal range, so equality is to rounding, not bit-exact
    _, gr_2 = _grads(g, x, gy, 2048.0)
    rel = (gr_2 - gr_a * 2).norm() / (gr_a * 2).norm()
    assert rel.item() < 2e-3, rel.item()


def test_full_geometry_inference_equals_training_forward(gen):
    g = gen
    x = torch.rand(1, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    g.train()
    y_t = g(x).detach().clone()
    g.eval()
    with torch.no_grad():
        y_e = g(x)
    g.train()
    assert torch.equal(y_t, y_e)      # rotating 3-workspace inference plan == saved-activation training plan


def test_training_reduces_the_loss():
    """System-level check of the fast path: RealESRNetStep (forward, L1, backward, GradScaler, fused Adam, EMA) on one
    fixed batch learns it -- the L1 loss falls by more than half in 60 steps and the EMA weights follow."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import RealESRNetStep
    import torch.nn.functional as F
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast").cuda().train()
    ema = R.EMA(g, 0.9)
    ema.register()
    opt = torch.optim.Adam(g.parameters(), 2e-4, (0.9, 0.99), fused=True)
    step = RealESRNetStep(g, ema, opt, torch.amp.GradScaler("cuda"), None)
    gen = torch.Generator(device="cuda").manual_seed(1)
    hr = F.interpolate(torch.rand(4, 3, 32, 32, device="cuda", generator=gen), size=(256, 256), mode="bicubic").clamp(0, 1)
    lr = F.interpolate(hr, scale_factor=0.25, mode="area")
    losses = [step(hr, lr).item() for _ in range(60)]
    assert all(map(lambda v: v == v, losses)), "NaN loss"
    first, last = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert last < 0.5 * first, (first, last)
    # EMA(0.9) after 60 steps sits close to the trained weights, far from the initial ones
    flat, shadow = g.flat_parameters(), ema._flat_shadow
    assert (flat - shadow).norm() < 0.5 * flat.norm() and torch.isfinite(shadow).all()


def test_flat_parameter_adam_matches_per_tensor_adam():
    """`Generator.flat_parameter()` mode (one Parameter aliasing the arena, gradient arena handed over as its `.grad`): three
    GradScaler + fused-Adam steps give the same weights as the per-tensor optimizer of the reference's script."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import RealESRNetStep

    def run(flat_mode):
        torch.manual_seed(11)
        g = R.Generator(3, 3, 4, n_blocks=2).cuda().train()
        params = [g.flat_parameter()] if flat_mode else g.parameters()
        opt = torch.optim.Adam(params, 2e-4, (0.9, 0.99), fused=True)
        scaler = torch.amp.GradScaler("cuda")
        step = RealESRNetStep(g, None, opt, scaler, None)
        gen = torch.Generator(device="cuda").manual_seed(5)
        losses = []
        for _ in range(3):
            lr = torch.rand(2, 3, 32, 48, device="cuda", generator=gen)
            hr = torch.rand(2, 3, 128, 192, device="cuda", generator=gen)
            losses.append(step(hr, lr).item())
        assert len(g.state_dict()) == len(list(g.parameters())) == 2 * (2 * 15 + 6)      # the alias is not a module parameter
        return g.flat_parameters().clone(), losses

    w_ref, l_ref = run(False)
    w_flat, l_flat = run(True)
    assert l_ref == l_flat
    assert torch.equal(w_ref, w_flat), (w_ref - w_flat).abs().max().item()


def test_exact16_at_full_geometry():
    """exact16 at BASELINE's full geometry (hi / lo tensors of 2.1 GB each at the HR end, offsets beyond 2^31 bytes), where the
    CPU oracle is out of reach: (a) inference at the bench batch (16 x 256^2) stays within fast mode's error class of the f16
    path on the same weights -- an addressing slip in the hi / lo pairs would be O(1); (b) a training step at 2 x 256^2 is
    deterministic, and its gradients agree with fast mode's to fast mode's accuracy while exact16's own linearity in the loss
    scale is fp32-class (the f16 path's is not)."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(3)
    gf = R.Generator(3, 3, 4, precision="fast").cuda()
    with torch.no_grad():
        gf.conv4.bias.add_(0.5)
    ge = R.Generator(3, 3, 4, precision="exact16")
    ge.load_state_dict(gf.state_dict())
    ge = ge.cuda()
    rng = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand(16, 3, 256, 256, device="cuda", generator=rng)
    with torch.no_grad():
        yf, ye = gf(x), ge(x)
    d = (yf - ye).abs()
    assert ye.shape == (16, 3, 1024, 1024) and torch.isfinite(ye).all()
    assert d.max().item() < 1e-2 and d.mean().item() < 5e-4, (d.max().item(), d.mean().item())
    del yf, ye, d
    torch.cuda.empty_cache()
    x2 = x[:2].contiguous()
    gy = torch.randn(2, 3, 1024, 1024, device="cuda", generator=rng) / 1024.0
    ge.train(); gf.train()
    y1, g1 = _grads(ge, x2, gy, 1024.0)
    y2, g2 = _grads(ge, x2, gy, 1024.0)
    assert torch.equal(y1, y2) and torch.equal(g1, g2)                       # deterministic
    _, g4 = _grads(ge, x2, gy, 4096.0)
    lin = ((g4 / 4.0 - g1).norm() / g1.norm()).item()
    assert lin < 1e-5, lin                                                    # linear in the loss scale to fp32 rounding
    _, gfast = _grads(gf, x2, gy, 1024.0)
    rel = ((gfast - g1).norm() / g1.norm()).item()
    assert rel < 0.1, rel                                                     # fast mode's distance (4-8 % per tensor at 23 blocks)


def test_tiled_4k_frame_two_tilings_agree():
    """BASELINE config 5 geometry (x2 model, 3840x2160 LR frame): the automatic plan (three 2160x1344 windows, one hipGraph for
    the whole frame) and an explicit 3 x 3 grid of 720x1280 tiles stitch the SAME 7680x4320 image bit for bit when the halo covers
    the receptive field (1-block trunk: ~21 LR pixels < halo 32) -- the whole-image pass itself is beyond the conv kernels'
    per-tensor pixel limit, which is why the tiler exists."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.tiling import TiledGenerator
    torch.manual_seed(5)
    g = R.Generator(3, 3, 2, precision="fast", n_blocks=1).cuda().eval()
    frame = torch.rand(1, 3, 2160, 3840, device="cuda")
    auto = TiledGenerator(g, halo=32, use_graph=True)
    tiles, wh, ww = auto.plan(1, 2160, 3840)
    assert len(tiles) == 3 and (wh, ww) == (2160, 1344)
    a = auto(frame).clone()
    assert torch.equal(auto(frame), a)                       # graph replay
    b = TiledGenerator(g, tile=(720, 1280), halo=32, use_graph=False)(frame)
    assert a.shape == b.shape == (1, 3, 4320, 7680) and torch.isfinite(a).all()
    assert torch.equal(a, b)


def test_discriminator_layer_mode_at_config4_size():
    """BASELINE config 4's per-GPU discriminator batch (16 x 256^2): the 256..512-channel layers' weight gradients in layer mode
    (one launch pair per layer, 2..8 pixel splits, work cut into eight XCD ranges) against the table-mode launch pairs -- equal to
    fp32 summation noise -- and the batched spectral-norm backward leaves every gradient finite."""
    import os
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(5)
    sd = R.Discriminator(precision="fast").state_dict()
    gen = torch.Generator(device="cuda").manual_seed(6)
    x = torch.rand(16, 3, 256, 256, device="cuda", generator=gen)
    gw = torch.randn(16, 1, 256, 256, device="cuda", generator=gen)

    def grads(table_mode):
        if table_mode:
            os.environ["RESR_WGRAD_NO_LAYER_MODE"] = "1"
        try:
            d = R.Discriminator(precision="fast")
            d.load_state_dict(sd)
            d = d.cuda().train()
            (d(x) * gw).sum().mul(64.0).backward()
            torch.cuda.synchronize()
            return {n: p.grad.clone() for n, p in d.named_parameters()}
        finally:
            os.environ.pop("RESR_WGRAD_NO_LAYER_MODE", None)
    layer, table = grads(False), grads(True)
    for n in layer:
        assert torch.isfinite(layer[n]).all(), n
        rel = ((layer[n] - table[n]).norm() / table[n].norm().clamp_min(1e-12)).item()
        assert rel < 2e-5, (n, rel)
