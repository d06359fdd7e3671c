"""Loss / loss-scale / gradient health over a few full-size train steps (same set-up as bench.py)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
from real_esrgan_pytorch_amd.train import RealESRNetStep
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--lr-size", type=int, default=128)
ap.add_argument("--per-tensor", action="store_true")
ap.add_argument("--precision", default="fast")
ap.add_argument("--smooth", action="store_true", help="smooth HR images instead of uniform noise")
a = ap.parse_args()
torch.manual_seed(0)
model = R.Generator(3, 3, 4, precision=a.precision).cuda().train()
opt = torch.optim.Adam(model.parameters() if a.per_tensor else [model.flat_parameter()], 2e-4, (0.9, 0.99), fused=True)
scaler = torch.amp.GradScaler("cuda") if a.precision == "fast" else None
g = torch.Generator(device="cuda").manual_seed(1234)
hr_edge = a.lr_size * 4
if a.smooth:
    hr = torch.nn.functional.interpolate(torch.rand(a.batch, 3, hr_edge // 16, hr_edge // 16, device="cuda", generator=g), size=(hr_edge, hr_edge), mode="bicubic").clamp(0, 1)
else:
    hr = torch.round(torch.rand(a.batch, 3, hr_edge, hr_edge, device="cuda", generator=g) * 255.0) / 255.0
lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode="area")
step = RealESRNetStep(model, None, opt, scaler, None)
w0 = model.flat_parameters().clone()
for i in range(a.steps):
    loss = step(hr, lr)
    if i % 5 == 0 or i == a.steps - 1:
        fg = model.flat_grad()
        print(f"step {i:3d} loss {loss.item():.5f} scale {scaler.get_scale() if scaler else 1:.0f} grad finite {bool(torch.isfinite(fg).all())} "
              f"|g|max {fg.abs().max().item():.3e} moved {(model.flat_parameters() - w0).abs().max().item():.3e} "
              f"sr mean {model(lr[:1]).mean().item():.4f}")
