"""Needs a TRACE BUILD of the library (the hooks are compiled out of the product build):
    python tools/build_variant.py trace -DRESR_TRACE=1 && RESR_LIB_PATH=$PWD/tools/ab/trace.so python tools/conv_timeline.py ...

Per-workgroup timeline of the fast-mode conv kernel (resr_debug_conv_trace): producer / consumer stamps in us."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--shape", default="160:32")
ap.add_argument("--wgs", default="0,1,8,9")
ap.add_argument("--stride", type=int, default=0)
ap.add_argument("--planar", action="store_true", help="chunk-planar input [C/32][N,H,W,32] as the generator uses")
a = ap.parse_args()
lib = L.lib()
cin, cout = map(int, a.shape.split(":"))
cout_pad = (cout + 31) // 32 * 32
mt = cout_pad // 32
gen = torch.Generator(device="cuda").manual_seed(1)
xs = a.stride or cin
x = (torch.rand(a.batch, a.res, a.res, xs, device="cuda", generator=gen) - 0.5).half()
y = torch.empty(a.batch, a.res, a.res, cout_pad, device="cuda", dtype=torch.float16)
w = ((torch.rand((cin // 32) * 9 * mt * 1024 + 8192, device="cuda", generator=gen) - 0.5) * 0.1).half()
d = L.ConvDesc(a.batch, a.res, a.res, cin, cin, xs, 0, cout, cout_pad, cout_pad, 0, 0, 0, L.RESR_F16, L.CONV_LRELU, 1, 1, 1, 1, 0.2)
if a.planar:
    x = (torch.rand(cin // 32, a.batch, a.res, a.res, 32, device="cuda", generator=gen) - 0.5).half()
    y = torch.empty(cout_pad // 32, a.batch, a.res, a.res, 32, device="cuda", dtype=torch.float16)
    d.in0_stride = 32; d.out_stride = 32
    d.in0_chunk_stride = a.batch * a.res * a.res * 32
    d.out_chunk_stride = a.batch * a.res * a.res * 32
def launch():
    L.check(lib.resr_conv3x3(C.byref(d), L.ptr(x), None, L.ptr(w), None, None, None, None, L.ptr(y), None, L.stream_ptr()))
for _ in range(3): launch()
torch.cuda.synchronize()
tr = torch.zeros(32 * 2 * 64, dtype=torch.int64, device="cuda")
lib.resr_debug_conv_trace(L.ptr(tr))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); launch(); e1.record(); torch.cuda.synchronize()
print(f"event-timed launch: {e0.elapsed_time(e1) * 1e3:.1f} us")
lib.resr_debug_conv_trace(None)
t = tr.cpu().view(32, 2, 64)
t0 = int(t[t > 0].min())
for wg in map(int, a.wgs.split(",")):
    for role, name in ((0, "producer"), (1, "consumer0")):
        v = [int(q) for q in t[wg, role] if q > 0]
        print(f"wg{wg} {name}: " + " ".join(f"{(q - t0) / 100:.2f}" for q in v))

import sys
if not all((t[wg] > 0).any() for wg in range(32)): sys.exit(0)
firsts = [min(int(q) for q in t[wg].flatten() if q > 0) - t0 for wg in range(32)]
lasts = [max(int(q) for q in t[wg].flatten() if q > 0) - t0 for wg in range(32)]
print("first stamp per wg (us):", " ".join(f"{v / 100:.1f}" for v in firsts))
print("last stamp per wg (us):", " ".join(f"{v / 100:.1f}" for v in lasts))
