"""Back-to-back launches of one dense-block convolution shape (chunk-planar operands, B x res^2), event-timed: what a change
to the conv kernel does to each of the generator's cout-32 / cout-64 passes without running a train step.
Usage: python tools/conv_loop.py [--batch 16] [--res 256] [--reps 300] [--shapes 64:32,96:32,128:32,160:32,192:64]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--shapes", default="64:32,96:32,128:32,160:32,192:64")
a = ap.parse_args()
lib = L.lib()
gen = torch.Generator(device="cuda").manual_seed(1)
for shape in a.shapes.split(","):
    cin, cout = map(int, shape.split(":"))
    mt = cout // 32
    px = a.batch * a.res * a.res
    x = (torch.randn(cin // 32, a.batch, a.res, a.res, 32, device="cuda", generator=gen) * 0.5).half()
    y = torch.empty(cout // 32, a.batch, a.res, a.res, 32, device="cuda", dtype=torch.float16)
    w = (torch.randn((cin // 32) * 9 * mt * 1024 + 8192, device="cuda", generator=gen) * 0.05).half()
    d = L.ConvDesc(a.batch, a.res, a.res, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16, L.CONV_LRELU, 1, 1, 1, 1, 0.2)
    d.in0_chunk_stride = px * 32
    d.out_chunk_stride = px * 32
    def launch():
        L.check(lib.resr_conv3x3(C.byref(d), L.ptr(x), None, L.ptr(w), None, None, None, None, L.ptr(y), None, L.stream_ptr()))
    for _ in range(a.reps):          # warm: reach the sustained power state
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps
    fl = 2.0 * 9 * cin * cout * px
    by = (cin + cout) * 2.0 * px
    print(f"{shape:>8s}  {us:8.1f} us/launch  {fl / us * 1e-6:7.1f} TFLOP/s  {by / us * 1e-6:6.2f} TB/s algorithmic")
