"""BASELINE config 3 (RealESRNet x4 L1 training, batch 32 of 256^2 HR tiles -> LR 64^2) on one MI355X."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
from real_esrgan_pytorch_amd.train import RealESRNetStep
from real_esrgan_pytorch_amd.degrade import Degrader
B = int(os.environ.get("B", "32")); steps = int(os.environ.get("STEPS", "20"))
torch.manual_seed(0)
g = R.Generator(3, 3, 4, precision=os.environ.get("PRECISION", "fast")).cuda().train()
ema = R.EMA(g, 0.999); ema.register()
GRAPH = os.environ.get("GRAPH") == "1"          # replay the step from one hipGraph (train.GraphedStep)
with torch.no_grad():
    g.conv4.bias.add_(0.5)                        # start inside the training-time clamp: the timed regime carries real gradients
opt = torch.optim.Adam([g.flat_parameter()], 2e-4, (0.9, 0.99), fused=True, capturable=GRAPH)
gen = torch.Generator(device="cuda").manual_seed(1)
base = torch.rand(B, 3, 16, 16, device="cuda", generator=gen)
hr = torch.nn.functional.interpolate(base, size=(256, 256), mode="bicubic").clamp(0, 1)
hr = torch.round((0.9 * hr + 0.1 * torch.rand(B, 3, 256, 256, device="cuda", generator=gen)) * 255) / 255
step = RealESRNetStep(g, ema, opt, torch.amp.GradScaler("cuda"), Degrader(batch=B, hr_size=256, upscale=4, crop=256, seed=0))
if GRAPH:
    from real_esrgan_pytorch_amd.train import GraphedStep
    step = GraphedStep(step)
for _ in range(8): step(hr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = step(hr)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(json.dumps({"config": f"3: RealESRNet x4 train, batch {B} of 256^2 HR tiles (LR 64^2)", "ms_per_step": round(dt * 1e3, 2),
                  "images_per_s": round(B / dt, 1), "tflops": round(B / dt * 0.4406, 1), "loss": float(loss)}))
