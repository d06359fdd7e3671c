import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import real_esrgan_pytorch_amd as R
from oracle import model_ref as M
nb = 6
for scale1 in (0.002, 0.05, 1.0):
    sd = M.init_generator_state(17, 3, 3, 4, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < nb}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    for k in ("conv1.weight", "conv1.bias"):
        sd[k] = sd[k] * scale1
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 24, 28, generator=gen)
    gw = torch.randn(2, 3, 96, 112, generator=gen)
    sdo = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.double().clone().requires_grad_(True)
    yo = M.generator_forward(xo, sdo, 4, nb)
    (yo * gw.double()).sum().backward()
    for plan, ls in ((0, 256.0), (0, 1.0), (0, 65536.0)):
        g = R.Generator(3, 3, 4, precision="exact16", n_blocks=nb, x2_plan=plan)
        g.load_state_dict(sd); g = g.cuda().train()
        xd = x.cuda().requires_grad_(True)
        y = g(xd); (y * gw.cuda()).sum().mul(ls).backward(); torch.cuda.synchronize()
        rel = lambda a, b: ((a.double() - b).norm() / b.norm().clamp_min(1e-30)).item()
        errs = {k: rel(p.grad.cpu() / ls, sdo[k].grad) for k, p in g.named_parameters()}
        groups = {}
        for k, e in errs.items():
            gk = k.split(".")[0] if not k.startswith("trunk") else "trunk." + k.split(".")[1]
            groups.setdefault(gk, []).append(e)
        print(f"conv1 x {scale1}, loss scale {ls}: gx {rel(xd.grad.cpu() / ls, xo.grad):.2e} " + " ".join(f"{k}:{max(v):.1e}" for k, v in groups.items()), flush=True)
