"""exact16 inference at 16 x 256^2 under several x2_plan values, alternated twice on one box (default: 59 = 40 stages per dense block,
the growth chunks against f16 weights, against 123 = + MX-fp8 correction stages on the pair chunks: 30 stage-equivalents).

    python tools/time_infer_plans.py [plan ...] [--batch N --size S]
"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import real_esrgan_pytorch_amd as R
torch.manual_seed(0)
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("plans", nargs="*", type=int, default=[59, 123])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
a = ap.parse_args()
x = torch.rand(a.batch, 3, a.size, a.size, device="cuda")
for rep in range(2):
    for plan in a.plans:
        g = R.Generator(3, 3, 4, precision="exact16", x2_plan=plan).cuda().eval()
        with torch.no_grad():
            for _ in range(3): g(x)
            torch.cuda.synchronize(); t = time.time()
            for _ in range(10): g(x)
            torch.cuda.synchronize(); dt = (time.time() - t) / 10
        print("plan", plan, "%.2f ms  %.1f images/s" % (dt * 1e3, a.batch / dt), flush=True)
        del g
