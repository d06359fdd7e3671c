import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import tests.gpu_util as U
L = U.L
def run(n, h, w, cin, cout, mask, dtype, reps=3):
    g = torch.Generator().manual_seed(1)
    x = U.quant(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    a = U.to_nhwc(x, dtype)
    packed = U.pack_conv(wt, dtype)
    d = L.ConvDesc(n, h, w, cin, cin, cin, 0, cout, cout, cout, 0, 0, 0, dtype, L.CONV_NO_BIAS, 1, 1, 1, 1, 0.2)
    ref = F.conv2d(x, U.quant(wt, dtype), None, padding=1)
    mk = None
    if mask:
        m = U.quant(torch.randn(n, cout, h, w, generator=g), dtype)
        mk = U.to_nhwc(m, dtype); d.mask_stride = cout; d.flags |= L.CONV_MASK
        ref = ref * torch.where(m > 0, 1.0, 0.2)
    for r in range(reps):
        out = torch.zeros(n, h, w, cout, dtype=U.tdtype(dtype), device="cuda")
        L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(a), None, L.ptr(packed), None, None, None, L.ptr(mk), L.ptr(out), None, L.stream_ptr()))
        torch.cuda.synchronize()
        e = (U.from_nhwc(out, cout) - ref).abs()
        print(f"n={n} {h}x{w} cin={cin} cout={cout} mask={mask} dtype={dtype}: per-image max err {[round(e[i].max().item(), 6) for i in range(n)]}", flush=True)
for dtype in (L.RESR_F32, L.RESR_F16):
    run(2, 128, 128, 64, 64, True, dtype)
    run(2, 132, 68, 64, 64, False, dtype)
    run(2, 80, 96, 64, 64, True, dtype, reps=1)
    run(4, 64, 64, 192, 64, False, dtype, reps=1)
