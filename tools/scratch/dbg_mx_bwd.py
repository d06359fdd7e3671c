import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tests.test_gpu_mx as T
n, h, w, nb = 8, 32, 32, 3
gen = torch.Generator().manual_seed(5)
x = torch.rand(n, 3, h, w, generator=gen); gw = torch.randn(n, 3, 4 * h, 4 * w, generator=gen)
res = {}
for plan in (0, 128, 2, 27, 155):
    g, sd, M = T._setup(nb, 11, plan)
    xd = x.cuda().requires_grad_(True)
    y = g.train()(xd); (y * gw.cuda()).sum().mul(1024.0).backward(); torch.cuda.synchronize()
    res[plan] = ({k: p.grad.double().cpu() for k, p in g.named_parameters()}, xd.grad.double().cpu())
def rel(a, b): return ((a - b).norm() / b.norm()).item()
for plan in (128, 2, 27, 155):
    r = sorted(rel(res[plan][0][k], res[0][0][k]) for k in res[0][0])
    print("plan", plan, "vs 0: median %.2e worst %.2e gx %.2e" % (r[len(r) // 2], r[-1], rel(res[plan][1], res[0][1])))
