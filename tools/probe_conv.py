"""Run individual conv3x3 / wgrad launches at the bench geometry (for rocprofv3 --pmc passes and A/B timing)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--shapes", default="64:32,160:32,192:64")
ap.add_argument("--wgrad", action="store_true")
ap.add_argument("--stride", type=int, default=0, help="input/output pixel stride in channels (0 = packed)")
a = ap.parse_args()
lib = L.lib()
gen = torch.Generator(device="cuda").manual_seed(1)
for sh in a.shapes.split(","):
    cin, cout = map(int, sh.split(":"))
    cout_pad = (cout + 31) // 32 * 32
    mt = cout_pad // 32
    xs = a.stride or cin
    ys = a.stride or cout_pad
    x = (torch.rand(a.batch, a.res, a.res, xs, device="cuda", generator=gen) - 0.5).half()
    y = torch.empty(a.batch, a.res, a.res, ys, device="cuda", dtype=torch.float16)
    w = ((torch.rand((cin // 32) * 9 * mt * 1024 + 8192, device="cuda", generator=gen) - 0.5) * 0.1).half()
    d = L.ConvDesc(a.batch, a.res, a.res, cin, cin, xs, 0, cout, cout_pad, ys, 0, 0, 0, L.RESR_F16, L.CONV_LRELU, 1, 1, 1, 1, 0.2)
    def launch():
        L.check(lib.resr_conv3x3(C.byref(d), L.ptr(x), None, L.ptr(w), None, None, None, None, L.ptr(y), None, L.stream_ptr()))
    launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): launch()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    print(f"conv {cin}->{cout} B={a.batch} {a.res}^2: {ms*1e3:.1f} us  {2*9*cin*cout*a.batch*a.res*a.res/ms/1e9:.0f} TFLOP/s", flush=True)
    if a.wgrad:
        g = (torch.rand(a.batch, a.res, a.res, cout_pad, device="cuda", generator=gen) - 0.5).half()
        splits = max(1, min(128, 768 // ((cin // 32) * mt)))
        wd = L.WgradDesc(a.batch, a.res, a.res, cin, cin, cin, 0, cin, cout, cout_pad, cout_pad, L.RESR_F16, 0, splits, 1.0)
        part = torch.empty(lib.resr_wgrad_partial_bytes(C.byref(wd)) // 4, device="cuda")
        dw = torch.empty(cout, cin, 3, 3, device="cuda"); db = torch.empty(cout, device="cuda")
        def lw():
            L.check(lib.resr_conv3x3_wgrad(C.byref(wd), L.ptr(x), None, L.ptr(g), L.ptr(part), L.ptr(dw), L.ptr(db), L.stream_ptr()))
        lw(); torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps): lw()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print(f"wgrad {cin}->{cout} splits={splits}: {ms*1e3:.1f} us  {2*9*cin*cout*a.batch*a.res*a.res/ms/1e9:.0f} TFLOP/s", flush=True)
