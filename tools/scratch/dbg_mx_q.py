import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tests.test_gpu_mx as T
from tests import gpu_util as U
L = U.L
name, cin, pair_ch, cout, n, h, w, kind = ("tail_conv", 64, 64, 64, 1, 33, 50, "tail")
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g)
wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
bias = torch.randn(cout, generator=g) * 0.1
xb, x_hi, x_lo = T._pair_q_planar(x, pair_ch)
plane = n * h * w * 32
packed, mx_off = T._pack_mx(U, wt)
out = torch.full((3, cout // 32, n, h, w, 32), -7.0, dtype=torch.float16, device="cuda")
d = L.ConvDesc(n, h, w, cin, cin, 32, 0, cout, cout, 32, 0, 0, 0, L.RESR_F16X2, L.CONV_MX_PAIRS | L.CONV_LRELU, 1.0, 1.0, 1.0, 1.0, 0.2)
d.in0_chunk_stride = d.out_chunk_stride = plane
d.in0_lo_offset, d.in0_q_offset, d.out_lo_offset, d.w_mx_offset, d.out_q_offset = 2 * plane, 4 * plane, 2 * plane, mx_off, 4 * plane
bias_d = bias.cuda()
L.check(L.lib().resr_conv3x3(C.byref(d), L.ptr(xb), None, L.ptr(packed), L.ptr(bias_d), None, None, None, L.ptr(out), None, L.stream_ptr()), "conv")
torch.cuda.synchronize()
def unplanar(t): return t.cpu().permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
hi_o, lo_o = unplanar(out[0]), unplanar(out[1])
qb = torch.from_numpy(out[2].cpu().contiguous().numpy().view(np.uint8).copy()).reshape(cout // 32, n, h, w, 64)
q_hi = qb[..., :32].permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
q_lo = qb[..., 32:].permute(1, 0, 4, 2, 3).reshape(n, cout, h, w)
want_hi, want_lo = T.bf8_bits(hi_o).reshape(n, cout, h, w), T.bf8_bits(lo_o).reshape(n, cout, h, w)
mis = (q_hi != want_hi)
print("mismatch frac hi", mis.float().mean().item(), "lo", (q_lo != want_lo).float().mean().item())
print("per channel:", [round(v, 2) for v in mis.float().mean((0, 2, 3)).tolist()])
print("per x:", [round(v, 2) for v in mis.float().mean((0, 1, 2)).tolist()])
print("per y:", [round(v, 2) for v in mis.float().mean((0, 1, 3)).tolist()])
print("pixel (0,0): got hi", q_hi[0, :, 0, 0].tolist()); print("want hi", want_hi[0, :, 0, 0].tolist())
print("raw record chunk0 px0:", qb[0, 0, 0, 0].tolist())
print("untouched (-7.0 pattern 0xc700):", (qb == 0xc7).float().mean().item(), (qb == 0x00).float().mean().item())
