"""GPU diagnostic: sweep generator fwd/bwd parity over shapes (strict mode) and localise gx errors."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
from oracle import model_ref as M

def run(upscale, nb, n, h, w, precision="strict"):
    sd = M.init_generator_state(11, 3, 3, upscale, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < nb}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    g = R.Generator(3, 3, upscale, precision=precision, n_blocks=nb); g.load_state_dict(sd); g = g.cuda()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen); gw = torch.randn(n, 3, h*upscale, w*upscale, generator=gen)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    (M.generator_forward(xo, sdo, upscale, nb) * gw).sum().backward()
    xd = x.cuda().requires_grad_(True)
    (g(xd) * gw.cuda()).sum().backward()
    e = (xd.grad.cpu() - xo.grad).abs()
    scale = xo.grad.abs().max().item()
    per_img = [(e[i].max().item() / scale) for i in range(n)]
    idx = (e == e.max()).nonzero()[0].tolist()
    worst = max(((p.grad.cpu() - sdo[k].grad).abs().max().item() / max(sdo[k].grad.abs().max().item(), 1e-9), k) for k, p in g.named_parameters())
    print(f"up={upscale} nb={nb} n={n} {h}x{w}: gx rel err per image {['%.2e' % v for v in per_img]} argmax {idx}; worst param {worst[1]} {worst[0]:.2e}", flush=True)

for n, h, w in [(1, 33, 17), (2, 32, 32), (2, 20, 24), (2, 33, 17), (3, 8, 8), (2, 8, 40)]:
    run(4, 1, n, h, w)
