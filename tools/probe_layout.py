"""Hypothesis test: chunk-planar ([32-channel chunk][pixel][32]) vs pixel-major input layout for a 64->32 conv,
using the two-segment input path (each segment = one 32-channel plane)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib; lib = L.lib()
B, res = 8, 256
gen = torch.Generator(device="cuda").manual_seed(1)
w = ((torch.rand(2 * 9 * 1024 + 8192, device="cuda", generator=gen) - 0.5) * 0.1).half()
def run(tag, d, a, b, y):
    def launch():
        L.check(lib.resr_conv3x3(C.byref(d), L.ptr(a), L.ptr(b), L.ptr(w), None, None, None, None, L.ptr(y), None, L.stream_ptr()))
    launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): launch()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{tag}: {ms*1e3:.1f} us  {2*9*64*32*B*res*res/ms/1e9:.0f} TFLOP/s", flush=True)
x64 = (torch.rand(B, res, res, 64, device="cuda", generator=gen) - 0.5).half()
x192 = (torch.rand(B, res, res, 192, device="cuda", generator=gen) - 0.5).half()
p0 = (torch.rand(B, res, res, 32, device="cuda", generator=gen) - 0.5).half()
p1 = (torch.rand(B, res, res, 32, device="cuda", generator=gen) - 0.5).half()
y32 = torch.empty(B, res, res, 32, device="cuda", dtype=torch.float16)
y192 = torch.empty(B, res, res, 192, device="cuda", dtype=torch.float16)
mk = lambda s0, s1, cin0, os_: L.ConvDesc(B, res, res, 64, cin0, s0, s1, 32, 32, os_, 0, 0, 0, L.RESR_F16, L.CONV_LRELU, 1, 1, 1, 1, 0.2)
for rep in range(2):
    run("pixel-major in[.,64]  out[.,32] ", mk(64, 0, 64, 32), x64, None, y32)
    run("pixel-major in[.,192] out[.,192]", mk(192, 0, 64, 192), x192, None, y192)
    run("chunk-planar 2x[.,32] out[.,32] ", mk(32, 32, 32, 32), p0, p1, y32)
