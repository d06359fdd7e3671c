"""-m gpu: the chained dense-block launches (the four cout-32 passes of a dense block as ONE persistent launch whose jobs
wait for each other through per-tile flags, conv3x3_ws.h CH) against the same passes as four launches.

The arithmetic is identical, only the scheduling differs, so forward output and every gradient must be BIT-equal; the
device-side health counters (flag polls that timed out, workgroups on an unexpected XCD) must stay 0.  Reference
semantics of the block: /root/reference/model.py:87-98."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(g, x, gw, no_chain):
    if no_chain:
        os.environ["RESR_CONV_NO_CHAIN"] = "1"
    else:
        os.environ.pop("RESR_CONV_NO_CHAIN", None)
    try:
        for p in g.parameters():
            p.grad = None
        xd = x.clone().requires_grad_(True)
        y = g(xd)
        (y * gw).sum().mul(256.0).backward()
        torch.cuda.synchronize()
        return y.detach().clone(), [p.grad.detach().clone() for p in g.parameters()], xd.grad.detach().clone()
    finally:
        os.environ.pop("RESR_CONV_NO_CHAIN", None)


# n (a multiple of 8: every XCD owns whole images), h, w (even), dense blocks: one workgroup per XCD, three images per XCD
# with ragged tiles, one tile per workgroup (16- and 8-row tiles),
# ragged last tiles, and several tiles per workgroup
# ... and BASELINE's headline geometry (16 x 256^2: eight tiles per workgroup and job)
CASES = [(8, 16, 16, 1), (24, 18, 34, 1), (8, 24, 40, 2), (8, 64, 64, 1), (16, 36, 70, 1), (16, 128, 128, 1), (32, 64, 64, 1), (16, 256, 256, 1)]


@pytest.mark.parametrize("precision", ["fast", "exact16"])
@pytest.mark.parametrize("n,h,w,n_blocks", CASES)
def test_chain_equals_separate_launches(n, h, w, n_blocks, precision):
    """fast and exact16 (hi/lo pairs: three stages per chunk, the dependent chunk = a job's last three stages)."""
    import real_esrgan_pytorch_amd as R
    L = R._lib
    torch.manual_seed(3)
    g = R.Generator(3, 3, 4, precision=precision, n_blocks=n_blocks).cuda().train()
    with torch.no_grad():
        g.conv4.bias.add_(0.5)
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(n, 3, h, w, device="cuda", generator=gen)
    gw = torch.randn(n, 3, 4 * h, 4 * w, device="cuda", generator=gen)
    y0, g0, gx0 = _run(g, x, gw, no_chain=True)
    for rep in range(3):   # scheduling differs from run to run: repeat
        y1, g1, gx1 = _run(g, x, gw, no_chain=False)
        assert torch.equal(y0, y1), f"forward differs (rep {rep}): {(y0 - y1).abs().max().item()}"
        assert torch.equal(gx0, gx1), f"input gradient differs (rep {rep})"
        for i, (a, b) in enumerate(zip(g0, g1)):
            assert torch.equal(a, b), f"gradient tensor {i} differs (rep {rep}): {(a - b).abs().max().item()}"
    err = int(L.lib().resr_debug_chain_errors())
    assert err == 0, f"chain health counters: polls timed out {err & 0xffffffff}, misplaced workgroups {err >> 32}"


@pytest.mark.parametrize("precision", ["fast", "exact16"])
def test_chain_is_taken(precision):
    """The profiling records show ONE conv launch for the passes of a block when chaining is on."""
    import ctypes as C
    import real_esrgan_pytorch_amd as R
    L = R._lib
    lib = L.lib()
    g = R.Generator(3, 3, 4, precision=precision, n_blocks=1).cuda().train()
    x = torch.rand(8, 3, 32, 32, device="cuda")

    def launches(no_chain):
        if no_chain:
            os.environ["RESR_CONV_NO_CHAIN"] = "1"
        else:
            os.environ.pop("RESR_CONV_NO_CHAIN", None)
        try:
            lib.resr_profile_begin()
            g(x)
            torch.cuda.synchronize()
            buf = (L.ProfEntry * 4096)()
            return int(lib.resr_profile_end(C.cast(buf, C.c_void_p), 4096))
        finally:
            os.environ.pop("RESR_CONV_NO_CHAIN", None)
    a, b = launches(True), launches(False)
    # three dense blocks per RRDB; on a launch this small conv5 joins the chain as two more jobs: five launches become one
    assert a - b == 3 * 4, (a, b)
    os.environ["RESR_CHAIN_CONV5"] = "0"   # ... and without it, four become one
    try:
        c = launches(False)
    finally:
        os.environ.pop("RESR_CHAIN_CONV5", None)
    assert a - c == 3 * 3, (a, c)


def test_chain_from_two_streams():
    """Chained launches issued from two streams take turns (a launch from another stream first waits -- stream-side, an event --
    for the previous owner's last chain; nothing is synchronised on the host): two generators stepping on their own streams
    give what they give alone."""
    import real_esrgan_pytorch_amd as R
    L = R._lib
    torch.manual_seed(5)
    gs = [R.Generator(3, 3, 4, precision="fast", n_blocks=1).cuda().train() for _ in range(2)]
    xs = [torch.rand(8, 3, 32, 48, device="cuda") for _ in range(2)]
    with torch.no_grad():
        ref = [g(x).clone() for g, x in zip(gs, xs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [None, None]
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]), torch.no_grad():
                outs[i] = gs[i](xs[i]).clone()
    torch.cuda.synchronize()
    for i in (0, 1):
        assert torch.equal(outs[i], ref[i])
    assert int(L.lib().resr_debug_chain_errors()) == 0


@pytest.mark.parametrize("precision", ["fast", "exact16"])
@pytest.mark.parametrize("n,h,w", [(8, 40, 48), (16, 128, 96)])
def test_chain_inference(n, h, w, precision):
    """eval() forward (plain LeakyReLU epilogue, rotating workspaces): chained == four launches, bit for bit."""
    import real_esrgan_pytorch_amd as R
    L = R._lib
    torch.manual_seed(9)
    g = R.Generator(3, 3, 4, precision=precision, n_blocks=2).cuda().eval()
    x = torch.rand(n, 3, h, w, device="cuda")
    os.environ["RESR_CONV_NO_CHAIN"] = "1"
    try:
        with torch.no_grad():
            y0 = g(x).clone()
    finally:
        os.environ.pop("RESR_CONV_NO_CHAIN", None)
    for _ in range(3):
        with torch.no_grad():
            y1 = g(x).clone()
        assert torch.equal(y0, y1)
    assert int(L.lib().resr_debug_chain_errors()) == 0


@pytest.mark.parametrize("n,h,w", [(8, 24, 32), (16, 70, 36), (3, 20, 24)])
def test_conv3x3_chain_entry_vs_torch(n, h, w):
    """resr_conv3x3_chain on a chunk-planar dense-block workspace against torch (model.py:90-93: out_k =
    leaky_relu(conv_k(cat(x, out_1 .. out_{k-1})))), sign words included; n = 3 takes the one-launch-per-job fallback."""
    import ctypes as C
    import torch.nn.functional as F
    from tests import gpu_util as U
    L = U.L
    lib = L.lib()
    g = torch.Generator().manual_seed(n * 100 + h)
    x = U.quant(torch.randn(n, 64, h, w, generator=g), L.RESR_F16)
    ws = torch.zeros(6, n, h, w, 32, dtype=torch.float16, device="cuda")   # planes [x0 x1 | o1 | o2 | o3 | o4]
    ws[:2] = x.reshape(n, 2, 32, h, w).permute(1, 0, 3, 4, 2).half().cuda()
    plane = n * h * w * 32
    descs = (L.ConvDesc * 4)()
    wts, biases, packed, bias_d, signs = [], [], [], [], []
    for k in range(4):
        cin = 64 + 32 * k
        wt = torch.randn(32, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(32, generator=g) * 0.1
        wts.append(wt); biases.append(b)
        packed.append(U.pack_conv(wt, L.RESR_F16)); bias_d.append(b.cuda())
        signs.append(torch.zeros((n, h, w, 1), dtype=torch.int32, device="cuda"))
        d = L.ConvDesc(n, h, w, cin, cin, 32, 0, 32, 32, 32, 0, 0, 0, L.RESR_F16, L.CONV_LRELU | L.CONV_WRITE_SIGNBITS,
                       1.0, 1.0, 1.0, 1.0, 0.2)
        d.in0_chunk_stride = plane
        descs[k] = d
    arr = lambda ptrs: (C.c_void_p * 4)(*ptrs)
    outs = [ws.data_ptr() + (2 + k) * plane * 2 for k in range(4)]
    # the chain state is the caller's: resr_conv3x3_chain_state_bytes of device memory, zero-filled once
    state = torch.zeros(int(lib.resr_conv3x3_chain_state_bytes(n, h, w)), dtype=torch.uint8, device="cuda")
    L.check(lib.resr_conv3x3_chain(4, descs, L.ptr(ws), None, arr([p.data_ptr() for p in packed]),
                                   arr([b.data_ptr() for b in bias_d]), None, arr(outs),
                                   arr([s.data_ptr() for s in signs]), L.ptr(state), state.numel(), L.stream_ptr()), "resr_conv3x3_chain")
    torch.cuda.synchronize()
    feats = [x]
    for k in range(4):   # the reference recursion on the values the kernel stored (f16 activations)
        ref = F.leaky_relu(F.conv2d(torch.cat(feats, 1), U.quant(wts[k], L.RESR_F16), biases[k], padding=1), 0.2)
        got = ws[2 + k].float().cpu().permute(0, 3, 1, 2)
        err = (got - ref).abs().max().item()
        assert err < 2e-2 * max(1.0, ref.abs().max().item()), (k, err)
        bits = ((signs[k].cpu().to(torch.int64) & 0xFFFFFFFF).unsqueeze(-1) >> torch.arange(32)) & 1
        assert torch.equal(bits.reshape(n, h, w, 32).permute(0, 3, 1, 2).bool(), got > 0), f"sign words of job {k}"
        feats.append(got)
    assert int(lib.resr_debug_chain_errors()) == 0


def test_chain_training_trajectory():
    """Ten optimiser steps (GradScaler, fused Adam over the flat arena, EMA) with chaining on and off from the same seeds: every
    loss and the final weights are bit-equal (tools/chain_soak.py runs the same comparison for hundreds of steps at BASELINE's
    geometries)."""
    import real_esrgan_pytorch_amd as R
    from real_esrgan_pytorch_amd.train import RealESRNetStep

    def run(no_chain):
        if no_chain:
            os.environ["RESR_CONV_NO_CHAIN"] = "1"
        else:
            os.environ.pop("RESR_CONV_NO_CHAIN", None)
        try:
            torch.manual_seed(0)
            g = R.Generator(3, 3, 4, precision="fast", n_blocks=2).cuda().train()
            ema = R.EMA(g, 0.999)
            ema.register()
            opt = torch.optim.Adam([g.flat_parameter()], 2e-4, (0.9, 0.99), fused=True)
            gen = torch.Generator(device="cuda").manual_seed(1)
            hr = torch.rand(8, 3, 128, 128, device="cuda", generator=gen)
            lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode="area")
            step = RealESRNetStep(g, ema, opt, torch.amp.GradScaler("cuda"), None)
            losses = [float(step(hr, lr)) for _ in range(10)]
            return losses, g.flat_parameter().detach().clone()
        finally:
            os.environ.pop("RESR_CONV_NO_CHAIN", None)
    l0, p0 = run(True)
    l1, p1 = run(False)
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)
    assert int(R._lib.lib().resr_debug_chain_errors()) == 0


def test_pinned_pipeline_experiment():
    """RESR_CHAIN_PIPE (experiment): the four growth convolutions as a pinned pipeline -- every workgroup of an
    XCD runs ONE job, walking the bands behind its predecessor's flags -- gives the same bits as four launches."""
    import real_esrgan_pytorch_amd as R
    L = R._lib
    torch.manual_seed(3)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=1).cuda().train()
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(16, 3, 256, 256, device="cuda", generator=gen)
    gw = torch.randn(16, 3, 1024, 1024, device="cuda", generator=gen)
    y0, g0, gx0 = _run(g, x, gw, no_chain=True)
    os.environ["RESR_CHAIN_PIPE"] = "5,7,9,11"
    try:
        for rep in range(2):
            y1, g1, gx1 = _run(g, x, gw, no_chain=False)
            assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
            for a, b in zip(g0, g1):
                assert torch.equal(a, b)
    finally:
        os.environ.pop("RESR_CHAIN_PIPE", None)
    assert int(L.lib().resr_debug_chain_errors()) == 0


def test_chain_next_to_a_kernel_that_holds_cus():
    """A chained launch needs all its workgroups resident; a kernel of ANOTHER stream that holds CUs for a while (what an RCCL
    all-reduce waiting for a peer looks like: data-parallel training overlaps them, train.DataParallel) only delays it: the
    polls wait (up to ~4 s), nothing times out, the results are bit-equal."""
    import real_esrgan_pytorch_amd as R
    L = R._lib
    lib = L.lib()
    torch.manual_seed(3)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=2).cuda().train()
    gen = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(16, 3, 128, 128, device="cuda", generator=gen)
    gw = torch.randn(16, 3, 512, 512, device="cuda", generator=gen)
    y0, g0, gx0 = _run(g, x, gw, no_chain=True)
    side = torch.cuda.Stream()
    for rep in range(3):
        # 24 workgroups x 150 KB of LDS: 24 CUs cannot take a conv workgroup for 30 ms (several whole passes)
        L.check(lib.resr_debug_occupy(24, 150 * 1024, 30000, side.cuda_stream), "resr_debug_occupy")
        y1, g1, gx1 = _run(g, x, gw, no_chain=False)
        assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
        for a, b in zip(g0, g1):
            assert torch.equal(a, b)
    torch.cuda.synchronize()
    assert int(lib.resr_chain_errors()) == 0 and int(lib.resr_debug_chain_errors()) == 0


def test_chain_state_too_small_falls_back():
    """A chain state that does not cover the geometry (or none at all) means one launch per job -- same results."""
    import ctypes as C
    from tests import gpu_util as U
    L = U.L
    lib = L.lib()
    n, h, w = 8, 32, 32
    g = torch.Generator().manual_seed(1)
    ws = torch.zeros(6, n, h, w, 32, dtype=torch.float16, device="cuda")
    ws[:2] = torch.randn(2, n, h, w, 32, generator=g).half().cuda()
    plane = n * h * w * 32
    descs = (L.ConvDesc * 4)()
    packed, bias_d = [], []
    for k in range(4):
        cin = 64 + 32 * k
        packed.append(U.pack_conv(torch.randn(32, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5, L.RESR_F16))
        bias_d.append((torch.randn(32, generator=g) * 0.1).cuda())
        d = L.ConvDesc(n, h, w, cin, cin, 32, 0, 32, 32, 32, 0, 0, 0, L.RESR_F16, L.CONV_LRELU, 1.0, 1.0, 1.0, 1.0, 0.2)
        d.in0_chunk_stride = plane
        descs[k] = d
    arr = lambda ptrs: (C.c_void_p * 4)(*ptrs)
    res = []
    for state in (torch.zeros(int(lib.resr_conv3x3_chain_state_bytes(n, h, w)), dtype=torch.uint8, device="cuda"),
                  torch.zeros(128, dtype=torch.uint8, device="cuda"), None):
        ws[2:] = 0
        outs = [ws.data_ptr() + (2 + k) * plane * 2 for k in range(4)]
        L.check(lib.resr_conv3x3_chain(4, descs, L.ptr(ws), None, arr([p.data_ptr() for p in packed]), arr([b.data_ptr() for b in bias_d]),
                                       None, arr(outs), None, L.ptr(state), 0 if state is None else state.numel(), L.stream_ptr()),
                "resr_conv3x3_chain")
        torch.cuda.synchronize()
        res.append(ws[2:].clone())
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert res[0].abs().sum().item() > 0


def test_chain_timeout_poisons_the_output_and_counts():
    """A chained launch whose workgroups cannot all become resident for longer than the poll time-out (here: a stand-in kernel
    holds 200 CUs for 9 s) must fail LOUDLY: the planes it could not wait for are read as NaNs, so the output is NaN (a
    training loss shows it at once), and the host-mapped counter -- read without any synchronisation -- is non-zero.
    In a subprocess: the error state is sticky for the process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, torch
sys.path.insert(0, %r)
import real_esrgan_pytorch_amd as R
L = R._lib; lib = L.lib()
torch.manual_seed(0)
g = R.Generator(3, 3, 4, precision="fast", n_blocks=1).cuda().eval()
x = torch.rand(8, 3, 64, 64, device="cuda")
with torch.no_grad():
    y_ok = g(x).clone()
torch.cuda.synchronize()
assert int(lib.resr_chain_errors()) == 0 and torch.isfinite(y_ok).all()
side = torch.cuda.Stream()
import time
L.check(lib.resr_debug_occupy(200, 150 * 1024, 9000000, side.cuda_stream), "occupy")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
with torch.no_grad():
    y = g(x)
e1.record()
torch.cuda.synchronize()
e = int(lib.resr_chain_errors())
print("errors", e & 0xffffffff, "nan", bool(torch.isnan(y).any()), "forward ms", e0.elapsed_time(e1))
assert (e & 0xffffffff) > 0 and torch.isnan(y).any()
try:
    L.chain_health()
except RuntimeError as err:
    print("raised:", str(err)[:60])
else:
    raise SystemExit("chain_health did not raise")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]


def test_chained_launches_inside_a_captured_graph():
    """Batch 8 (n % 8 == 0: the dense blocks run as chained launches) captured into a hipGraph and replayed: under capture the
    launcher neither waits for, records, nor takes ownership of its per-device event (none of that would mean anything at replay
    time, csrc/conv3x3_ws.hip); replays are bit-equal to the eager pass, an EAGER chained pass from another stream afterwards
    still orders itself against the last eager owner, and no poll ever times out."""
    import real_esrgan_pytorch_amd as R
    lib = R._lib.lib()
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=2).cuda().eval()
    x = torch.rand(8, 3, 64, 64, device="cuda")
    with torch.no_grad():
        y_eager = g(x).clone()
        for _ in range(2):
            g(x)
        torch.cuda.synchronize()
        static_x = x.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_y = g(static_x)
        for i in range(3):
            static_x.copy_(x if i != 1 else x.flip(0))
            graph.replay()
            torch.cuda.synchronize()
            want = y_eager if i != 1 else y_eager.flip(0)
            assert torch.equal(static_y, want), i
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            y_side = g(x)                                     # eager, another stream: waits for the last EAGER owner's event
        side.synchronize()
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(y_side, y_eager) and torch.equal(static_y, y_eager)
    assert int(lib.resr_debug_chain_errors()) == 0


def test_chain_grid_headroom_knob_keeps_results():
    """RESR_CHAIN_CUS_PER_XCD (read per call): the chained launches on at most k workgroups per XCD -- room for a co-resident
    collective -- produce bit-equal outputs and gradients (tile ownership follows a workgroup's XCD ticket, whatever the grid)."""
    import real_esrgan_pytorch_amd as R
    torch.manual_seed(0)
    g = R.Generator(3, 3, 4, precision="fast", n_blocks=2).cuda().train()
    x = torch.rand(8, 3, 64, 64, device="cuda")
    res = []
    for k in (None, "31", "7"):
        if k:
            os.environ["RESR_CHAIN_CUS_PER_XCD"] = k
        try:
            g.zero_grad(set_to_none=True)
            y = g(x)
            y.square().sum().mul(64.0).backward()
            torch.cuda.synchronize()
            res.append((y.detach().clone(), g.flat_grad().clone()))
        finally:
            os.environ.pop("RESR_CHAIN_CUS_PER_XCD", None)
    for y, gr in res[1:]:
        assert torch.equal(y, res[0][0]) and torch.equal(gr, res[0][1])
    assert int(R._lib.lib().resr_debug_chain_errors()) == 0
