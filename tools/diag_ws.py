import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import tests.gpu_util as U
L = U.L
def run(n, h, w, cin, cout, dtype=L.RESR_F16):
    g = torch.Generator().manual_seed(1)
    x = U.quant(torch.randn(n, cin, h, w, generator=g), dtype)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    a = U.to_nhwc(x, dtype)
    packed = U.pack_conv(wt, dtype)
    d = L.ConvDesc(n, h, w, cin, cin, cin, 0, cout, cout, cout, 0, 0, 0, dtype, L.CONV_NO_BIAS, 1, 1, 1, 1, 0.2)
    ref = F.conv2d(x, U.quant(wt, dtype), None, padding=1)
    out = torch.zeros(n, h, w, cout, dtype=U.tdtype(dtype), device="cuda")
    L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(a), None, L.ptr(packed), None, None, None, None, L.ptr(out), None, L.stream_ptr()))
    torch.cuda.synchronize()
    o = U.from_nhwc(out, cout)
    e = (o - ref).abs()
    print(f"n={n} {h}x{w} cin={cin} cout={cout}: max err {e.max().item():.4f}")
    bad = e > 0.05
    print(" bad frac", bad.float().mean().item(), " by channel:", [round(bad[:, c].float().mean().item(), 2) for c in range(cout)])
    print(" bad by row:", [round(bad[0, :, r].float().mean().item(), 2) for r in range(min(h, 20))])
    print(" bad by col:", [round(bad[0, :, :, c].float().mean().item(), 2) for c in range(min(w, 36))])
    # does output channel c equal reference channel perm?
    for c in range(0, cout, 4):
        d2 = (o[0, c].unsqueeze(0) - ref[0]).abs().flatten(1).max(1).values
        print("  out ch", c, "closest ref ch", int(d2.argmin()), float(d2.min()))
run(1, 16, 32, 32, 32)
run(1, 40, 40, 64, 64)
