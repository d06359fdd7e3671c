"""exact16 against strict (f32 MFMA) on one small generator: forward, input gradient, every weight gradient (backward order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import real_esrgan_pytorch_amd as R  # noqa: E402

n, h, w, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
gs = R.Generator(3, 3, 4, precision="strict", n_blocks=nb).cuda().train()
with torch.no_grad():
    gs.conv4.bias.add_(0.5)
    for p in gs.parameters():
        if p.ndim == 1:
            p.add_(0.02 * torch.randn_like(p))
gx = R.Generator(3, 3, 4, precision="exact16", n_blocks=nb).cuda().train()
gx.load_state_dict(gs.state_dict())
gen = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand(n, 3, h, w, device="cuda", generator=gen)
gw = torch.randn(n, 3, 4 * h, 4 * w, device="cuda", generator=gen)
res = {}
for name, g, sc in (("strict", gs, 1.0), ("exact16", gx, 1024.0)):
    xd = x.clone().requires_grad_(True)
    y = g(xd)
    (y * gw).sum().mul(sc).backward()
    torch.cuda.synchronize()
    res[name] = (y.detach(), xd.grad / sc, {k: p.grad / sc for k, p in g.named_parameters()})
ys, gxs, gs_ = res["strict"]
yx, gxx, gx_ = res["exact16"]
print("forward max abs", (ys - yx).abs().max().item())
rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
print("gx rel", rel(gxx, gxs))
for k in reversed(list(gs_)):
    e = rel(gx_[k], gs_[k])
    if e > 1e-3 or "-v" in sys.argv:
        print(f"{k:40s} {e:.3e}")
