"""ContentLoss(detached=False): gradient wrt every VGG pre-activation against the CPU oracle's autograd (layer by layer)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_ref as M  # noqa: E402
from real_esrgan_pytorch_amd.content_loss import ContentLoss  # noqa: E402

nodes = ["features.2", "features.7", "features.16", "features.25", "features.34"]
weights = [0.1, 0.1, 1.0, 1.0, 1.0]
mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
aliasing = "--no-alias" not in sys.argv
precision = "exact16" if "--exact16" in sys.argv else "strict"
seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 6
torch.manual_seed(4)
cl = ContentLoss(nodes, mean, std, precision=precision, inplace_relu_aliasing=aliasing, detached=False).cuda()
sd = {k: v.detach().cpu() for k, v in cl.state_dict().items() if k.startswith("features.")}
gen = torch.Generator().manual_seed(seed)
sr, hr = torch.rand(2, 3, 64, 48, generator=gen), torch.rand(2, 3, 64, 48, generator=gen)

# oracle with every pre-activation retained
m, s = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)
wanted = {int(k.split(".")[1]) for k in nodes}
last = max(wanted)


def feats(x, pres):
    x = (x - m) / s
    out, idx = {}, 0
    for v in M.VGG19_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2)
            idx += 1
            continue
        pre = F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1)
        if pres is not None:
            pre.retain_grad()
            pres[idx] = pre
        x = F.relu(pre)
        if idx in wanted:
            out[f"features.{idx}"] = pre if (not aliasing or idx == last) else x
        if idx == last:
            break
        idx += 2
    return out


sro = sr.clone().requires_grad_(True)
pres = {}
a, b = feats(sro, pres), feats(hr, None)
sum(w * F.l1_loss(a[k], b[k]) for w, k in zip(weights, nodes)).backward()

cl._debug_grads = {}
srd = sr.cuda().requires_grad_(True)
got = cl(srd, hr.cuda())
scale = 1.0 if precision == "strict" else 4096.0
(sum(w * l for w, l in zip(weights, got)) * scale).backward()
torch.cuda.synchronize()
rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30)).item()
for idx in sorted(pres, reverse=True):
    ref, mine = pres[idx].grad, cl._debug_grads[idx] / scale
    d = (mine - ref).abs()
    print(f"features.{idx:2d} pre-activation gradient rel {rel(mine, ref):.3e}  max abs {d.max().item():.3e} (ref max {ref.abs().max().item():.3e})  "
          f"elements off by > 1e-3 of max: {(d > 1e-3 * ref.abs().max()).sum().item()} of {d.numel()}")
print("d/d sr rel", rel(srd.grad.cpu() / scale, sro.grad))
