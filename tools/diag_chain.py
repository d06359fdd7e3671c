"""resr_conv3x3_chain on one dense-block workspace against torch, job by job: where (which job / rows / columns / channels) does
a chained launch differ?  python tools/diag_chain.py [n h w]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U  # noqa: E402

L = U.L
lib = L.lib()
n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 24, 32)
g = torch.Generator().manual_seed(n * 100 + h)
x = U.quant(torch.randn(n, 64, h, w, generator=g), L.RESR_F16)
ws = torch.zeros(6, n, h, w, 32, dtype=torch.float16, device="cuda")
ws[:2] = x.reshape(n, 2, 32, h, w).permute(1, 0, 3, 4, 2).half().cuda()
plane = n * h * w * 32
descs = (L.ConvDesc * 4)()
wts, biases, packed, bias_d = [], [], [], []
for k in range(4):
    cin = 64 + 32 * k
    wt = torch.randn(32, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(32, generator=g) * 0.1
    wts.append(wt); biases.append(b)
    packed.append(U.pack_conv(wt, L.RESR_F16)); bias_d.append(b.cuda())
    d = L.ConvDesc(n, h, w, cin, cin, 32, 0, 32, 32, 32, 0, 0, 0, L.RESR_F16, L.CONV_LRELU, 1.0, 1.0, 1.0, 1.0, 0.2)
    d.in0_chunk_stride = plane
    descs[k] = d
arr = lambda ptrs: (C.c_void_p * 4)(*ptrs)
outs = [ws.data_ptr() + (2 + k) * plane * 2 for k in range(4)]
state = torch.zeros(int(lib.resr_conv3x3_chain_state_bytes(n, h, w)), dtype=torch.uint8, device="cuda")
L.check(lib.resr_conv3x3_chain(4, descs, L.ptr(ws), None, arr([p.data_ptr() for p in packed]), arr([b.data_ptr() for b in bias_d]), None, arr(outs),
                               None, L.ptr(state), state.numel(), L.stream_ptr()), "resr_conv3x3_chain")
torch.cuda.synchronize()
feats = [x]
for k in range(4):
    ref = F.leaky_relu(F.conv2d(torch.cat(feats, 1), U.quant(wts[k], L.RESR_F16), biases[k], padding=1), 0.2)
    got = ws[2 + k].float().cpu().permute(0, 3, 1, 2)
    bad = (got - ref).abs() > 2e-2 * max(1.0, ref.abs().max().item())
    print(f"job {k}: max err {(got - ref).abs().max().item():.3e}, wrong {bad.float().mean().item():.3f}; by image {[round(v, 2) for v in bad.float().mean((1, 2, 3)).tolist()]}")
    print("   by row", [round(v, 2) for v in bad.float().mean((0, 1, 3)).tolist()])
    print("   by channel group of 8", [round(v, 2) for v in bad.float().mean((0, 2, 3)).reshape(4, 8).mean(1).tolist()])
    feats.append(ref)          # continue from the correct values: isolate each job
print("chain errors", int(lib.resr_debug_chain_errors()))
