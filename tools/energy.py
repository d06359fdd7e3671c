"""Energy per FLOP of the three hot kernels of the headline train step, each ALONE in a back-to-back loop for >= `--seconds`
(the sustained power state, not a burst), with the board's hwmon power and shader clock sampled every 50 ms (bench.PowerSampler):

    chain32   the chained cout-32 launch of a dense block, training form (conv1..4, LeakyReLU + sign words), B x res^2
    cout64    the closing convolution 192 -> 64 (conv3x3_ws_kernel<f16,2,4,4>)
    wgrad     wgrad_quad_kernel + its slab reduction: the batched launch pair of one RRDB (three dense blocks, 78 products)

The step sits on the board's power cap: what limits it is joules, so the kernel to work on is the one with the
worst pJ/FLOP, not the one with the most milliseconds.  Columns: us / launch, algorithmic TFLOP/s and TB/s, average W and MHz over
the loop, pJ per algorithmic FLOP -- total board energy, and above the idle draw measured at the start.

    python tools/energy.py [--batch 16] [--res 256] [--seconds 2.0] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import real_esrgan_pytorch_amd as R  # noqa: E402
from bench import PowerSampler  # noqa: E402

L = R._lib
lib = L.lib()
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--only", default="")
ap.add_argument("--json", default="")
a = ap.parse_args()
n, h, w = a.batch, a.res, a.res
px = n * h * w
gen = torch.Generator(device="cuda").manual_seed(1)


def packed_weights(cin, cout):
    mt = cout // 32
    return (torch.randn((cin // 32) * 9 * mt * 1024 + 8192, device="cuda", generator=gen) * 0.05).half()


def make_chain():
    ws = torch.zeros(6, n, h, w, 32, dtype=torch.float16, device="cuda")            # planes [x0 x1 | o1 | o2 | o3 | o4]
    ws[:2] = (torch.randn(2, n, h, w, 32, device="cuda", generator=gen) * 0.5).half()
    plane = px * 32
    descs = (L.ConvDesc * 4)()
    packed, bias, signs = [], [], []
    for k in range(4):
        cin = 64 + 32 * k
        packed.append(packed_weights(cin, 32))
        bias.append(torch.zeros(32, device="cuda"))
        signs.append(torch.zeros((n, h, w, 1), dtype=torch.int32, device="cuda"))
        d = L.ConvDesc(n, h, w, cin, cin, 32, 0, 32, 32, 32, 0, 0, 0, L.RESR_F16, L.CONV_LRELU | L.CONV_WRITE_SIGNBITS, 1.0, 1.0, 1.0, 1.0, 0.2)
        d.in0_chunk_stride = plane
        descs[k] = d
    arr = lambda ptrs: (C.c_void_p * 4)(*ptrs)   # noqa: E731
    outs = arr([ws.data_ptr() + (2 + k) * plane * 2 for k in range(4)])
    pk, bs, sg = arr([p.data_ptr() for p in packed]), arr([b.data_ptr() for b in bias]), arr([s.data_ptr() for s in signs])
    state = torch.zeros(int(lib.resr_conv3x3_chain_state_bytes(n, h, w)), dtype=torch.uint8, device="cuda")
    keep = (ws, packed, bias, signs, state)

    def launch():
        L.check(lib.resr_conv3x3_chain(4, descs, L.ptr(ws), None, pk, bs, None, outs, sg, L.ptr(state), state.numel(), L.stream_ptr()), "chain")
    flop = sum(2.0 * 9 * (64 + 32 * k) * 32 * px for k in range(4))
    byts = sum(((64 + 32 * k) + 32) * 2.0 * px + 4.0 * px for k in range(4))
    return launch, flop, byts, keep


def make_cout64():
    cin, cout = 192, 64
    x = (torch.randn(cin // 32, n, h, w, 32, device="cuda", generator=gen) * 0.5).half()
    y = torch.empty(cout // 32, n, h, w, 32, device="cuda", dtype=torch.float16)
    wt = packed_weights(cin, cout)
    d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16, 0, 1, 1, 1, 1, 0.2)
    d.in0_chunk_stride = px * 32
    d.out_chunk_stride = px * 32

    def launch():
        L.check(lib.resr_conv3x3(C.byref(d), L.ptr(x), None, L.ptr(wt), None, None, None, None, L.ptr(y), None, L.stream_ptr()), "conv")
    return launch, 2.0 * 9 * cin * cout * px, (cin + cout) * 2.0 * px, (x, y, wt)


def make_wgrad():
    # Exactly the launch pair the step issues per RRDB: the five convolutions of three dense blocks as ONE batch (78 products = 20
    # quad jobs x 24 pixel splits by the generator's own rule, generator.hip splits_for; the jobs of a split share an XCD's L2),
    # on chunk-planar operands ([6][N,H,W,32] per block, as the generator keeps them) -- resr_debug_wgrad_dense_blocks.
    nb = 3
    xs = [(torch.rand(6, n, h, w, 32, device="cuda", generator=gen) - 0.5).half() for _ in range(nb)]
    gs = [(torch.rand(6, n, h, w, 32, device="cuda", generator=gen) - 0.5).half() for _ in range(nb)]
    quads = (26 * nb + 3) // 4
    splits = 512 // quads
    if splits >= 16:
        splits &= ~7
    splits = max(1, min(splits, 256, (n * ((h + 7) // 8) * ((w + 31) // 32)) // 2))
    partial = torch.empty(26 * nb * splits * (9 * 1024 + 32), device="cuda")
    dw = torch.empty(nb * 26624 * 9, device="cuda")
    xp = (C.c_void_p * nb)(*[t.data_ptr() for t in xs])
    gp = (C.c_void_p * nb)(*[t.data_ptr() for t in gs])

    def launch():
        L.check(lib.resr_debug_wgrad_dense_blocks(nb, xp, gp, n, h, w, splits, L.ptr(partial), partial.numel() * 4, L.ptr(dw), L.stream_ptr()), "wgrad")
    return launch, 2.0 * 9 * 26624 * nb * px, nb * 12 * 64.0 * px, (xs, gs, partial, dw)


def sample_idle(seconds=1.0):
    torch.cuda.synchronize()
    s = PowerSampler()
    with s:
        time.sleep(seconds)
    return s.summary()


def run(name, make):
    launch, flop, byts, keep = make()
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 50
    warm = max(20, int(0.7 / per))
    reps = max(50, int(a.seconds / per))
    for _ in range(warm):                     # reach the sustained power state before sampling
        launch()
    torch.cuda.synchronize()
    s = PowerSampler()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with s:
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    p = s.summary() or {}
    row = {"kernel": name, "us_per_launch": round(us, 2), "tflops": round(flop / us * 1e-6, 1), "tb_per_s_algorithmic": round(byts / us * 1e-6, 2),
           "avg_w": p.get("avg_w"), "sclk_mhz": p.get("sclk_mhz_avg"), "reps": reps}
    if p.get("avg_w"):
        row["pj_per_flop_total"] = round(p["avg_w"] * us * 1e-6 / flop * 1e12, 3)
        if IDLE and IDLE.get("avg_w"):
            row["pj_per_flop_above_idle"] = round((p["avg_w"] - IDLE["avg_w"]) * us * 1e-6 / flop * 1e12, 3)
    del keep
    torch.cuda.empty_cache()
    return row


IDLE = sample_idle()
rows = []
for name, make in (("chain32 (conv1..4 of a dense block, training form)", make_chain), ("cout64 (192 -> 64)", make_cout64),
                   ("wgrad_quad + reduce (one RRDB: 78 products)", make_wgrad)):
    if a.only and a.only not in name:
        continue
    rows.append(run(name, make))
    time.sleep(0.5)
print(f"batch {n} x {h}^2, loops of >= {a.seconds} s; idle board: {IDLE}")
for r in rows:
    print(f"{r['kernel']:52s} {r['us_per_launch']:9.1f} us  {r['tflops']:7.1f} TFLOP/s  {r['tb_per_s_algorithmic']:5.2f} TB/s  {r['avg_w']} W  {r['sclk_mhz']} MHz  "
          f"{r.get('pj_per_flop_total')} pJ/FLOP total, {r.get('pj_per_flop_above_idle')} above idle")
print("chain errors", int(lib.resr_debug_chain_errors()))
if a.json:
    with open(a.json, "w") as f:
        json.dump({"batch": n, "res": h, "idle": IDLE, "rows": rows}, f, indent=1)
