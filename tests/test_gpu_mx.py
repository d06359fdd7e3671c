"""-m gpu: exact16 with MX-fp8 correction stages (RESR_CONV_MX_PAIRS, RESR_X2_PLAN_MX_INFER -- DESIGN.md section 2, round 6).

A pair chunk of an inference forward takes ONE f16 stage (x_hi W0) and ONE stage of v_mfma_scale_f32_32x32x64_f8f6f4 on unscaled bf8
(e5m2) operands for both 2^-12-weighted corrections: B = the pixel's q record [bf8(x_hi) x 32 | bf8(x_lo) x 32] written by the producing
epilogue, A = [bf8(W1) | bf8(W2)] from resr_pack_weights_mx.  Reference arithmetic: /root/reference/model.py:87-98,255-272 (fp32 on the
CPU, inference.py:52-53).

What is held:
  * one pass against the float64 evaluation of EXACTLY its arithmetic (the operand roundings are the only error source of the MFMA
    path) -- which pins the instruction's operand layout, the packer's fragment order, the stage map and the q addressing;
  * the q records an epilogue / the input layout kernel writes, byte for byte, against the f16 -> e5m2 rounding of the stored hi / lo
    values (v_cvt_scalef32_pk_bf8_f16 at unit scale == round the 16-bit pattern to its upper byte, nearest even);
  * the whole 23-block forward against the fp32 CPU oracle: gate 2e-4 at the reference's init, with the dense
    weights x 4 and on trained weights; chained and unchained launches; tools/mx_infer_sim.py predicts 0.8-1.2e-4.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def U():
    from tests import gpu_util
    return gpu_util


def bf8_bits(h16: torch.Tensor) -> torch.Tensor:
    """f16 tensor -> uint8 e5m2 patterns, round to nearest even (csrc/pack.hip f16_bits_to_bf8)."""
    b = torch.from_numpy(h16.cpu().contiguous().numpy().view(np.uint16).astype(np.int64))
    return (((b + 0x7f + ((b >> 8) & 1)) >> 8) & 0xff).to(torch.uint8)


def bf8_value(h16: torch.Tensor) -> torch.Tensor:
    """The value an e5m2 byte stands for, as float64 (its pattern is the upper byte of an f16)."""
    bits = (bf8_bits(h16).to(torch.int32) << 8).numpy().astype(np.uint16)
    return torch.from_numpy(bits.view(np.float16).astype(np.float64)).reshape(h16.shape)


def _planar(v, n, c, h, w):
    return v.reshape(n, c // 32, 32, h, w).permute(1, 0, 3, 4, 2).contiguous()


def _pair_q_planar(t, pair_ch):
    """[N,C,H,W] fp32 -> chunk-planar buffer [3][C/32][N,H,W,32] f16-sized: hi planes, lo planes, q planes (64 B per pixel and chunk:
    32 bytes bf8(hi) | 32 bytes bf8(lo)); + the f16 hi / scaled lo tensors."""
    n, c, h, w = t.shape
    hi = t.half()
    lo = ((t - hi.float()) * 4096.0).half()
    q = torch.cat([_planar(bf8_bits(hi).reshape(n, c, h, w), n, c, h, w), _planar(bf8_bits(lo).reshape(n, c, h, w), n, c, h, w)], -1)   # [C/32][N,H,W,64] bytes
    buf = torch.zeros(3, c // 32, n, h, w, 32, dtype=torch.float16)
    buf[0], buf[1] = _planar(hi, n, c, h, w), _planar(lo, n, c, h, w)
    buf[2] = torch.from_numpy(q.numpy().view(np.float16).copy()).reshape(c // 32, n, h, w, 32)
    if pair_ch < c:
        buf[1, pair_ch // 32:] = 777.0       # lo / q planes of the single chunks: poison, must not be read
        buf[2, pair_ch // 32:] = 777.0
    return buf.cuda(), hi, lo


def _pack_mx(U, wt):
    """f16x2 blocks + the MX region behind them (one pack table for both), and the byte offset of the region."""
    L = U.L
    cout, cin = wt.shape[:2]
    mt, nck = (cout + 31) // 32, (cin + 31) // 32
    chunks = (L.PackChunk * nck)()
    for ck in range(nck):
        chunks[ck] = L.PackChunk(0, ck * 9 * mt * 1024, cout, cin, 0, cout, ck * 32, min(32, cin - ck * 32), mt, 0, 1.0, 0, None)
    table = torch.frombuffer(bytearray(bytes(chunks)), dtype=torch.uint8).cuda()
    arena = wt.reshape(-1).float().cuda()
    plain = nck * 9 * mt * 1024
    mx_off = (plain * 6 + 16384 + 255) // 256 * 256
    packed = torch.zeros(mx_off + plain * 2 + 16384, dtype=torch.uint8, device="cuda")
    L.check(L.lib().resr_pack_weights(L.ptr(table), nck, L.ptr(arena), L.ptr(packed), L.RESR_F16X2, L.stream_ptr()), "resr_pack_weights")
    L.check(L.lib().resr_pack_weights_mx(L.ptr(table), nck, L.ptr(arena), C.c_void_p(packed.data_ptr() + mx_off), L.stream_ptr()), "resr_pack_weights_mx")
    return packed, mx_off


def _mx_reference(x_hi, x_lo, wt, bias, pair_ch, up=False):
    """float64 evaluation of the pass's arithmetic: x_hi W0 + bf8(x_hi) bf8(W1) + bf8(x_lo) bf8(W2) on the pair channels, x W0 on the
    single ones (RESR_CONV_SINGLE_W16), everything times 2^-12 at the end (the weights carry 2^12, x_lo is stored times 2^12)."""
    t = wt.float() * 4096.0
    w0 = t.half()
    w1 = (t - w0.float()).half()
    w2 = (w0.float() / 4096.0).half()
    xh, xl = x_hi.double(), x_lo
    if up:
        xh, xl = F.interpolate(xh, scale_factor=2, mode="nearest"), F.interpolate(xl.float(), scale_factor=2, mode="nearest").half()
        x_hi = F.interpolate(x_hi.float(), scale_factor=2, mode="nearest").half()
    p = pair_ch
    acc = F.conv2d(xh, w0.double(), None, padding=1)
    acc = acc + F.conv2d(bf8_value(x_hi[:, :p]), bf8_value(w1[:, :p]), None, padding=1) + F.conv2d(bf8_value(xl[:, :p]), bf8_value(w2[:, :p]), None, padding=1)
    return acc / 4096.0 + bias.double().view(1, -1, 1, 1)


MX_CASES = [
    # name, cin, pair_ch, cout, n, h, w, kind
    ("growth_conv3", 128, 64, 32, 2, 40, 36, "growth"),       # conv1..4 of a dense block: 2 pair chunks + single f16 growth planes, LeakyReLU, single output
    ("closing_conv5", 192, 64, 64, 2, 36, 70, "closing"),     # residual, pair output + q output
    ("closing_small_rows8", 192, 64, 64, 1, 12, 40, "closing"),
    ("tail_conv", 64, 64, 64, 1, 33, 50, "tail"),             # HR tail: all pairs, LeakyReLU, pair output + q output
    ("tail_up", 64, 64, 64, 1, 36, 40, "tail_up"),            # ... with the nearest x2 gather of the q records
    ("first_conv", 32, 32, 64, 2, 21, 37, "tail"),            # conv1: one (padded) pair chunk
    ("rows16", 64, 64, 64, 8, 200, 200, "tail"),              # 16-row tiles, several tiles per workgroup, ragged edges
    ("rows16_growth", 160, 64, 32, 8, 200, 200, "growth"),
]


@pytest.mark.parametrize("case", MX_CASES, ids=[c[0] for c in MX_CASES])
def test_mx_pass_matches_its_arithmetic(U, case, diag_dir):
    L = U.L
    name, cin, pair_ch, cout, n, h, w, kind = case
    g = torch.Generator().manual_seed(len(name) + cin)
    up = kind == "tail_up"
    hs, ws = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, cin, hs, ws, generator=g)
    if name == "first_conv":
        x[:, 3:] = 0.0          # channels beyond the real three are padding
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    xb, x_hi, x_lo = _pair_q_planar(x, pair_ch)
    plane_in, plane = n * hs * ws * 32, n * h * w * 32
    packed, mx_off = _pack_mx(U, wt)
    out = torch.full((3, cout // 32, n, h, w, 32), -7.0, dtype=torch.float16, device="cuda")
    d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16X2, L.CONV_MX_PAIRS, 1.0, 1.0, 1.0, 1.0, 0.2)
    d.in0_chunk_stride, d.out_chunk_stride = plane_in, plane
    d.in0_lo_offset, d.in0_q_offset = (cin // 32) * plane_in, 2 * (cin // 32) * plane_in
    d.out_lo_offset = (cout // 32) * plane
    d.w_mx_offset = mx_off
    if pair_ch < cin:
        d.x2_pair_chunks = pair_ch // 32
        d.flags |= L.CONV_SINGLE_W16
    ref = _mx_reference(x_hi, x_lo, wt, bias, pair_ch, up)
    res0 = None
    if kind == "growth":
        d.flags |= L.CONV_LRELU | L.CONV_OUT_SINGLE
        ref = F.leaky_relu(ref, 0.2)
    elif kind == "closing":
        r0 = torch.randn(n, cout, h, w, generator=g)
        res0, r_hi, r_lo = _pair_q_planar(r0, cout)
        d.res0_stride, d.res0_chunk_stride, d.s0, d.t0, d.res0_lo_offset = 32, plane, 0.2, 1.0, (cout // 32) * plane
        d.out_q_offset = 2 * (cout // 32) * plane
        ref = ref * 0.2 + (r_hi.double() + r_lo.double() / 4096.0)
    else:
        d.flags |= L.CONV_LRELU | (L.CONV_UPSAMPLE_IN if up else 0)
        d.out_q_offset = 2 * (cout // 32) * plane
        ref = F.leaky_relu(ref, 0.2)
    bias_d = bias.cuda()
    L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(xb), None, L.ptr(packed), L.ptr(bias_d), L.ptr(res0), None, None,
                                 L.ptr(out), None, L.stream_ptr()), "resr_conv3x3")
    torch.cuda.synchronize()

    def unplanar(t):
        return t.cpu().permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
    rep = {"ref_absmax": ref.abs().max().item()}
    if kind == "growth":
        assert (out[1] == -7.0).all() and (out[2] == -7.0).all(), "a single f16 output leaves the lo and q tensors alone"
        got = unplanar(out[0]).double()
        bound = ref.abs() * (2.0 ** -11 * 1.01) + 2e-6
    else:
        hi_o, lo_o = unplanar(out[0]), unplanar(out[1])
        got = hi_o.double() + lo_o.double() / 4096.0
        bound = torch.full_like(ref, 2e-6 * max(1.0, ref.abs().max().item()))
        # the q tensor: per pixel and chunk 32 bytes bf8(hi) then 32 bytes bf8(lo), exactly the e5m2 rounding of what was stored
        qb = torch.from_numpy(out[2].cpu().contiguous().numpy().view(np.uint8).copy()).reshape(cout // 32, n, h, w, 64)
        q_hi = qb[..., :32].permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
        q_lo = qb[..., 32:].permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
        rep["q_hi_mismatch"] = int((q_hi != bf8_bits(hi_o).reshape(n, cout, h, w)).sum())
        rep["q_lo_mismatch"] = int((q_lo != bf8_bits(lo_o).reshape(n, cout, h, w)).sum())
    err = (got - ref).abs()
    rep.update({"max_abs_err": err.max().item(), "worst_over_bound": (err / bound).max().item()})
    with open(os.path.join(diag_dir, f"mx_conv_{name}.json"), "w") as f:
        json.dump(rep, f)
    assert (err <= bound).all(), (name, rep)
    assert rep.get("q_hi_mismatch", 0) == 0 and rep.get("q_lo_mismatch", 0) == 0, rep


def test_mx_flag_needs_its_operands(U):
    """RESR_CONV_MX_PAIRS without the q offset / MX blocks, on another dtype, or with a training epilogue is an argument error."""
    L = U.L
    n, h, w, cin, cout = 1, 16, 32, 64, 64
    plane = n * h * w * 32
    xb = torch.zeros(3, 2, n, h, w, 32, dtype=torch.float16, device="cuda")
    out = torch.zeros(3, 2, n, h, w, 32, dtype=torch.float16, device="cuda")
    packed, mx_off = _pack_mx(U, torch.zeros(cout, cin, 3, 3))

    def desc(**kw):
        d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16X2, L.CONV_MX_PAIRS, 1.0, 1.0, 1.0, 1.0, 0.2)
        d.in0_chunk_stride = d.out_chunk_stride = plane
        d.in0_lo_offset, d.in0_q_offset, d.out_lo_offset, d.w_mx_offset = 2 * plane, 4 * plane, 2 * plane, mx_off
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    def run(d, aux=None):
        return L.lib().resr_conv3x3(C.byref(d), L.ptr(xb), None, L.ptr(packed), None, None, None, None, L.ptr(out), L.ptr(aux), L.stream_ptr())
    assert run(desc()) == 0
    assert run(desc(in0_q_offset=0)) == -1
    assert run(desc(w_mx_offset=0)) == -1
    assert run(desc(dtype=L.RESR_F16)) == -1
    assert run(desc(flags=L.CONV_MX_PAIRS | L.CONV_LRELU | L.CONV_WRITE_SIGNBITS), aux=torch.zeros(n * h * w * 2, dtype=torch.int32, device="cuda")) == -1
    assert run(desc(flags=0, out_q_offset=4 * plane)) == 0         # (any exact16 pass with a lean epilogue may EMIT a q tensor: the pass behind it decides)
    assert run(desc(flags=L.CONV_OUT_SINGLE, out_q_offset=4 * plane)) == -1   # ... but not next to a single-f16 output
    torch.cuda.synchronize()


def _setup(n_blocks, seed, x2_plan, wscale=1.0):
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    sd = M.init_generator_state(seed, 3, 3, 4, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < n_blocks}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    if wscale != 1.0:
        sd = {k: (v * wscale if k.endswith(".weight") and ".rdb" in k else v) for k, v in sd.items()}
    g = R.Generator(3, 3, 4, precision="exact16", n_blocks=n_blocks, x2_plan=x2_plan)
    g.load_state_dict(sd)
    return g.cuda(), sd, M


@pytest.mark.parametrize("n,h,w,wscale", [(1, 24, 24, 1.0), (8, 24, 24, 1.0), (2, 24, 28, 4.0)])
def test_mx_inference_forward_vs_oracle(n, h, w, wscale, diag_dir):
    """23 blocks, eval, x2_plan 97 = bits 0 + 5 + 6 (30 stage-equivalents per dense block) against the fp32 CPU oracle, its float64
    evaluation and round 5's 40-stage plan (33); weights at the reference's init and with the dense-block weights x 4; 8 x 32^2 runs the
    8 x 24^2 runs the dense blocks as chained launches (six jobs: the closing convolution's halves emit the q records), the others as
    separate launches on ragged tiles.  (2 x 72 x 100 -- 16-row tiles -- and 8 x 32^2 at x 4 were measured when the stage was built:
    1.12e-4 / 1.13e-4 and 1.14e-4, gpurun_out/mx_infer_*.json of round 6; the per-pass cases above cover the 16-row shapes.)"""
    from real_esrgan_pytorch_amd import _lib as L
    gm, sd, M = _setup(23, 11, 97, wscale)
    g33, _, _ = _setup(23, 11, 33, wscale)
    x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ym = gm.eval()(x.cuda()).cpu()
        y33 = g33.eval()(x.cuda()).cpu()
    yo = M.generator_forward(x, sd, 4, 23)     # (the fp32 CPU path IS the gate's reference; its float64 evaluation differs from it by < 2e-6 here)
    rep = {"mx_vs_f32_oracle": (ym - yo).abs().max().item(), "plan33_vs_f32_oracle": (y33 - yo).abs().max().item(), "mx_vs_plan33": (ym - y33).abs().max().item(),
           "mx_mean_abs_vs_f32_oracle": (ym - yo).abs().mean().item(), "frac_unclamped": ((yo > 0) & (yo < 1)).float().mean().item()}
    with open(os.path.join(diag_dir, f"mx_infer_{n}x{h}x{w}_w{wscale}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["mx_vs_f32_oracle"] < 2e-4, rep
    assert rep["mx_vs_plan33"] > 0 and rep["plan33_vs_f32_oracle"] < 5e-5, rep     # the MX stages really ran; the reference plan is where it was
    assert int(L.lib().resr_debug_chain_errors()) == 0


def test_mx_inference_chained_equals_separate_launches(monkeypatch):
    """The chained dense-block launches of the MX plan against one launch per convolution: bit for bit (16 x 64^2: six-job chains on
    8-row tiles; 16 x 128^2: four-job chains on 16-row tiles + the closing convolution's own launch)."""
    for n, h, w in ((16, 64, 64), (16, 128, 128)):
        g, _, _ = _setup(2, 11, 97)
        x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(3)).cuda()
        with torch.no_grad():
            y_chain = g.eval()(x).clone()
            monkeypatch.setenv("RESR_CONV_NO_CHAIN", "1")
            y_sep = g(x).clone()
            monkeypatch.delenv("RESR_CONV_NO_CHAIN")
        assert torch.equal(y_chain, y_sep), (n, h, w, (y_chain - y_sep).abs().max().item())


def test_mx_inference_after_training_steps(diag_dir):
    """The same gate on weights that have left the init: 60 RealESRNet steps (fast mode) on one fixed batch, then MX inference."""
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
    ge = R.Generator(3, 3, 4, precision="exact16", x2_plan=97)
    ge.load_state_dict(sd)
    with torch.no_grad():
        ym = ge.cuda().eval()(lr[:1]).cpu()
    yo = M.generator_forward(lr[:1].cpu(), sd, 4, 23)
    rep = {"mx_vs_f32_oracle": (ym - yo).abs().max().item()}
    with open(os.path.join(diag_dir, "mx_infer_trained.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["mx_vs_f32_oracle"] < 2e-4, rep


def test_mx_plan_is_inference_only_and_needs_its_prerequisites():
    """A training forward ignores the bit (same bits as the plan without it); the bit without bits 0 + 5 is refused."""
    import real_esrgan_pytorch_amd as R
    with pytest.raises(ValueError):
        R.Generator(3, 3, 4, precision="exact16", x2_plan=64)
    with pytest.raises(ValueError):
        R.Generator(3, 3, 4, precision="exact16", x2_plan=65)
    torch.manual_seed(1)
    ga = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=123).cuda().train()
    gb = R.Generator(3, 3, 4, precision="exact16", n_blocks=2, x2_plan=59).cuda().train()
    gb.load_state_dict(ga.state_dict())
    x = torch.rand(2, 3, 24, 40, device="cuda")
    ya, yb = ga(x), gb(x)
    assert torch.equal(ya, yb)
    ya.sum().backward()
    yb.sum().backward()
    for (ka, pa), (kb, pb) in zip(ga.named_parameters(), gb.named_parameters()):
        assert torch.equal(pa.grad, pb.grad), ka


@pytest.mark.parametrize("n,h,w,n_blocks,seed", [(1, 24, 24, 23, 11), (8, 32, 32, 3, 11), (2, 33, 17, 2, 13), (16, 32, 32, 2, 7)])
def test_mx_backward_plan_gradients(n, h, w, n_blocks, seed, diag_dir):
    """x2_plan 27 + 128 (RESR_X2_PLAN_MX_BWD): the dense blocks' backward-data passes read every gradient chunk as a pair on one f16 + one
    MX stage.  Every gradient tensor against the float64 evaluation of the oracle AND against the all-pairs plan on the same device
    (the training forward is the same bits, hence the same LeakyReLU masks: the distance between the two is the plan's own effect).
    Gate of VERDICT round 5, item 1b: every tensor <= 5e-4 against the all-pairs plan (tools/precision_ladder_sim.py, rung "bf8x4096 on
    both operands": 0.8-1.2e-4; round 5's plan 27: 2-5e-4).  8 x 32^2 and 16 x 32^2 run the mirrored passes as chained
    launches (g_x's halves inside the chain), 2 x 33 x 17 as separate launches on ragged tiles.  (Seed 12 at full depth and 16 x 64^2
    were measured when the plan was built: worst 1.7e-4 / 2.4e-4, the same tensors as plan 27's.)"""
    gm, sd, M = _setup(n_blocks, seed, 27 + 128)
    g27, _, _ = _setup(n_blocks, seed, 27)
    g0, _, _ = _setup(n_blocks, seed, 0)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen)
    gw = torch.randn(n, 3, 4 * h, 4 * w, generator=gen)
    # (the all-pairs plan on the same device is the reference here: its own distance to the float64 oracle -- 6e-6, or a mask flip of the
    # shared forward pass -- is held by test_gpu_generator.py and test_gpu_x2_plan.py)

    def run(model):
        xd = x.cuda().requires_grad_(True)
        y = model.train()(xd)
        (y * gw.cuda()).sum().mul(1024.0).backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), {name: p.grad.cpu().double() / 1024.0 for name, p in model.named_parameters()}, xd.grad.cpu().double() / 1024.0
    y, gr, gx = run(gm)
    y27, gr27, _ = run(g27)
    y0, gr0, gx0 = run(g0)
    assert torch.equal(y, y0) and torch.equal(y27, y0), "the training forward does not depend on the plan"

    def rel(got, ref):
        return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    between = {name: rel(gr[name], gr0[name]) for name in gr}
    between27 = {name: rel(gr27[name], gr0[name]) for name in gr}
    worst_b = max(between, key=between.get)
    vals = sorted(between.values())
    rep = {"worst_vs_all_pairs_plan": between[worst_b], "worst_vs_all_pairs_tensor": worst_b, "median_vs_all_pairs_plan": vals[len(vals) // 2],
           "plan27_worst_vs_all_pairs_plan": max(between27.values()), "gx_vs_all_pairs": rel(gx, gx0)}
    with open(os.path.join(diag_dir, f"mx_bwd_{n}x{h}x{w}_{n_blocks}_{seed}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["worst_vs_all_pairs_plan"] < 5e-4 and rep["gx_vs_all_pairs"] < 2e-4, rep
    assert rep["worst_vs_all_pairs_plan"] > 1e-5, "the plan does not seem to be active (gradients at the all-pairs level)"


def test_mx_backward_chained_equals_separate_launches(monkeypatch):
    """The mirrored chained launches of the MX backward plan against one launch per pass: every gradient bit for bit."""
    import real_esrgan_pytorch_amd as R
    for n, s in ((16, 64), (8, 128)):
        g, _, _ = _setup(2, 11, 27 + 128)
        x = torch.rand(n, 3, s, s, generator=torch.Generator().manual_seed(3)).cuda()
        gw = torch.randn(n, 3, 4 * s, 4 * s, generator=torch.Generator().manual_seed(4)).cuda()

        def grads():
            g.zero_grad(set_to_none=True)
            (g.train()(x) * gw).sum().mul(256.0).backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in g.named_parameters()}
        a = grads()
        monkeypatch.setenv("RESR_CONV_NO_CHAIN", "1")
        b = grads()
        monkeypatch.delenv("RESR_CONV_NO_CHAIN")
        assert all(torch.equal(a[k], b[k]) for k in a), (n, s)


def test_exact16_forward_with_f16_backward(diag_dir):
    """x2_plan bit 8 (opt-in): exact16's all-pairs forward -- the same bits as any exact16 training forward, hence
    reference-exact LeakyReLU masks -- followed by fast mode's backward pass on the hi tensors.  Against the all-pairs plan under a dense
    random cotangent the emulation reads worst 1.6-2.0e-3 / median 0.9-1.1e-3 where fast mode reads 5-8e-2 / 3-4e-2 (tools/precision_ladder_sim.py,
    "fwd exact; bwd: g f16 all, W f16"): most of fast mode's gradient error is its own forward's mask flips."""
    n, h, w, nb = 8, 32, 32, 3
    gh, sd, M = _setup(nb, 11, 256)
    g0, _, _ = _setup(nb, 11, 0)
    import real_esrgan_pytorch_amd as R
    gf = R.Generator(3, 3, 4, precision="fast", n_blocks=nb)
    gf.load_state_dict(sd)
    gf = gf.cuda()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen)
    gw = torch.randn(n, 3, 4 * h, 4 * w, generator=gen)

    def run(model):
        xd = x.cuda().requires_grad_(True)
        y = model.train()(xd)
        (y * gw.cuda()).sum().mul(1024.0).backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), {name: p.grad.cpu().double() / 1024.0 for name, p in model.named_parameters()}
    yh, grh = run(gh)
    y0, gr0 = run(g0)
    yf, grf = run(gf)
    assert torch.equal(yh, y0), "the forward pass is exact16's"

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    eh = sorted(rel(grh[k], gr0[k]) for k in gr0)
    ef = sorted(rel(grf[k], gr0[k]) for k in gr0)
    rep = {"hybrid_median": eh[len(eh) // 2], "hybrid_worst": eh[-1], "fast_median": ef[len(ef) // 2], "fast_worst": ef[-1]}
    with open(os.path.join(diag_dir, "hybrid_f16_backward.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["hybrid_worst"] < 5e-3 and rep["hybrid_median"] < 2e-3, rep
    assert rep["hybrid_median"] < 0.25 * rep["fast_median"], rep


@pytest.mark.parametrize("n,h,w,n_blocks,seed", [(8, 32, 32, 3, 11), (2, 33, 17, 2, 13), (1, 24, 24, 23, 11)])
def test_mx_weight_gradient_jobs(n, h, w, n_blocks, seed, diag_dir):
    """x2_plan bit 9 (RESR_X2_PLAN_MX_WGRAD, on top of 27 + 128): the stream chunks' correction tap-products (x_hi, g_lo) + (x_lo, g_hi) of
    every dense-block weight gradient as ONE MX job -- 8-bit transpose reads of the staged q records, K = 32 pixels twice per
    v_mfma_scale_f32_32x32x64_f8f6f4 --, which also sums g_lo's share of the bias from the bf8 bytes of its G fragments (against the plan without
    the bit, whose bias takes the f16 lo tensor of the SAME gradient planes, a bias moves by the bf8 rounding of a 2^-12-weighted term: 1.4-2.3e-5 measured, gate 5e-5;
    leaving the share out, or summing the wrong bytes, reads 2^-12 ~ 2.4e-4).  The training forward (which now also emits the
    stream's q tensor) stays the all-pairs plan's bit for bit; every gradient tensor within 5e-4 of the all-pairs plan (VERDICT round 5,
    item 1c); against the plan without the bit the MX jobs move a tensor by ~1e-5 (emulation: weight gradients alone 2-3e-5 worst) --
    more where they restore the (x_hi, g_lo) term that plan bit 1 drops for conv1..conv4."""
    gw9, sd, M = _setup(n_blocks, seed, 27 + 128 + 512)
    g7, _, _ = _setup(n_blocks, seed, 27 + 128)
    g0, _, _ = _setup(n_blocks, seed, 0)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen)
    gwt = torch.randn(n, 3, 4 * h, 4 * w, generator=gen)

    def run(model):
        xd = x.cuda().requires_grad_(True)
        y = model.train()(xd)
        (y * gwt.cuda()).sum().mul(1024.0).backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), {name: p.grad.cpu().double() / 1024.0 for name, p in model.named_parameters()}, xd.grad.cpu().double() / 1024.0
    y9, gr9, gx9 = run(gw9)
    y7, gr7, gx7 = run(g7)
    y0, gr0, gx0 = run(g0)
    assert torch.equal(y9, y0) and torch.equal(gx9, gx7), "forward and backward-data do not depend on the weight-gradient jobs"

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    b9 = {k: rel(gr9[k], gr0[k]) for k in gr0}
    b7 = {k: rel(gr7[k], gr0[k]) for k in gr0}
    d97 = {k: rel(gr9[k], gr7[k]) for k in gr0}
    trunk = [k for k in gr0 if ".rdb" in k]
    other = [k for k in gr0 if ".rdb" not in k]
    rep = {"worst_vs_all_pairs_plan": max(b9.values()), "worst_tensor": max(b9, key=b9.get), "without_the_bit_worst": max(b7.values()),
           "median_vs_all_pairs_plan": sorted(b9.values())[len(b9) // 2], "without_the_bit_median": sorted(b7.values())[len(b7) // 2],
           "moved_by_the_mx_jobs_worst": max(d97[k] for k in trunk), "worst_bias_vs_all_pairs": max(b9[k] for k in trunk if k.endswith(".bias")),
           "bias_moved_by_the_mx_jobs_worst": max(d97[k] for k in trunk if k.endswith(".bias"))}
    with open(os.path.join(diag_dir, f"mx_wgrad_{n}x{h}x{w}_{n_blocks}_{seed}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert all(torch.equal(gr9[k], gr7[k]) for k in other), "the HR tail / conv1 / conv2 keep their f16 tap-products"
    assert 0 < rep["moved_by_the_mx_jobs_worst"] < 3e-4, rep
    assert rep["bias_moved_by_the_mx_jobs_worst"] < 5e-5, rep
    assert rep["worst_vs_all_pairs_plan"] < 5e-4 and rep["worst_bias_vs_all_pairs"] < 2e-4, rep


@pytest.mark.parametrize("n,h,w,n_blocks,seed", [(2, 33, 17, 1, 13), (4, 32, 32, 2, 11)])
def test_mx_tail_plan_gradients(n, h, w, n_blocks, seed, diag_dir):
    """x2_plan bit 10 (RESR_X2_PLAN_MX_TAIL, on top of 27 + 128 + 512): conv3, conv4 and upsampling2 -- the 4x-resolution tail -- the way the
    dense blocks run: the training forward emits the q tensors of u1, u2 and c3 (the LeakyReLU + sign-word epilogues), the layout pass and the
    two masked tail passes those of g4 / gA / gB, the three tail passes that read them take one f16 + one MX stage per chunk, and the three
    weight gradients their correction tap-products as MX jobs (upsampling2's X is read through the nearest-neighbour upsampling).  The
    training forward stays the all-pairs plan's bit for bit; every gradient tensor within 5e-4 of the all-pairs plan; against the plan
    without the bit only what lies UPSTREAM of the tail's passes and the tail's own three weights may move."""
    gt, sd, M = _setup(n_blocks, seed, 27 + 128 + 512 + 1024)
    g9, _, _ = _setup(n_blocks, seed, 27 + 128 + 512)
    g0, _, _ = _setup(n_blocks, seed, 0)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen)
    gwt = torch.randn(n, 3, 4 * h, 4 * w, generator=gen)

    def run(model):
        xd = x.cuda().requires_grad_(True)
        y = model.train()(xd)
        (y * gwt.cuda()).sum().mul(1024.0).backward()
        torch.cuda.synchronize()
        return y.detach().cpu(), {name: p.grad.cpu().double() / 1024.0 for name, p in model.named_parameters()}, xd.grad.cpu().double() / 1024.0
    yt, grt, gxt = run(gt)
    y9, gr9, gx9 = run(g9)
    y0, gr0, gx0 = run(g0)
    assert torch.equal(yt, y0), "the training forward does not depend on the plan"

    def rel(a, b):
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    bt = {k: rel(grt[k], gr0[k]) for k in gr0}
    b9 = {k: rel(gr9[k], gr0[k]) for k in gr0}
    moved = {k: rel(grt[k], gr9[k]) for k in gr0}
    rep = {"worst_vs_all_pairs_plan": max(bt.values()), "worst_tensor": max(bt, key=bt.get), "without_the_bit_worst": max(b9.values()),
           "median_vs_all_pairs_plan": sorted(bt.values())[len(bt) // 2], "gx_vs_all_pairs": rel(gxt, gx0),
           "tail_weights_vs_all_pairs": {k: bt[k] for k in bt if k.split(".")[0] in ("conv3", "conv4", "upsampling2")},
           "moved_by_the_bit_worst": max(moved.values())}
    with open(os.path.join(diag_dir, f"mx_tail_{n}x{h}x{w}_{n_blocks}_{seed}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert 0 < rep["moved_by_the_bit_worst"] < 5e-4, rep       # the bit is active
    assert rep["worst_vs_all_pairs_plan"] < 5e-4 and rep["gx_vs_all_pairs"] < 2e-4, rep
