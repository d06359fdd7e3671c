"""RealESRGAN (GAN) train step on one MI355X: BASELINE config 4 per-GPU share (batch 16, HR 256^2 crops) and the
headline geometry (HR 1024^2).  Parity-test configuration, not the headline bench."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
from real_esrgan_pytorch_amd.train import RealESRGANStep
from real_esrgan_pytorch_amd.degrade import Degrader

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--hr", type=int, default=256)
ap.add_argument("--tile", type=int, default=400)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--content", action="store_true")
ap.add_argument("--precision", default="fast", choices=["fast", "exact16", "strict"])
ap.add_argument("--graph", action="store_true", help="replay the step from one hipGraph (train.GraphedStep)")
a = ap.parse_args()
torch.manual_seed(0)
g = R.Generator(3, 3, 4, precision=a.precision).cuda().train()
d = R.Discriminator(precision=a.precision).cuda().train()
ema = R.EMA(g, 0.999); ema.register()
ap_flat = not os.environ.get("RESR_PER_TENSOR_ADAM")
go = torch.optim.Adam([g.flat_parameter()] if ap_flat else g.parameters(), 1e-4, (0.9, 0.99), fused=True, capturable=a.graph)
do = torch.optim.Adam([d.flat_parameter()] if ap_flat else d.parameters(), 1e-4, (0.9, 0.99), fused=True, capturable=a.graph)
cl = R.ContentLoss(["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], precision=a.precision).cuda() if a.content else None
deg = Degrader(batch=a.batch, hr_size=a.tile, upscale=4, crop=a.hr, seed=0)
step = RealESRGANStep(g, d, ema, go, do, torch.amp.GradScaler("cuda"), deg, content_criterion=cl)
if a.graph:
    from real_esrgan_pytorch_amd.train import GraphedStep
    step = GraphedStep(step)
hr = torch.round(torch.rand(a.batch, 3, a.tile, a.tile, device="cuda") * 255) / 255
for _ in range(6 if a.graph else 2): out = step(hr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): out = step(hr)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"config": f"RealESRGAN step, batch {a.batch}, HR tile {a.tile}^2 -> crop {a.hr}^2, content={a.content}",
                  "ms_per_step": round(dt * 1e3, 2), "images_per_s": round(a.batch / dt, 2),
                  "losses": {k: round(float(v), 5) for k, v in out.items()}}))
