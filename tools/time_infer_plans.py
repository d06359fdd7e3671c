"""exact16 inference at 16 x 256^2 under two x2_plan values, alternated twice on one box (default: 27 = 50 stages per dense block against
59 = 40 stages, the growth chunks against f16 weights).

    python tools/time_infer_plans.py
"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import real_esrgan_pytorch_amd as R
torch.manual_seed(0)
x = torch.rand(16, 3, 256, 256, device="cuda")
for rep in range(2):
    for plan in (27, 59):
        g = R.Generator(3, 3, 4, precision="exact16", x2_plan=plan).cuda().eval()
        with torch.no_grad():
            for _ in range(3): g(x)
            torch.cuda.synchronize(); t = time.time()
            for _ in range(10): g(x)
            torch.cuda.synchronize(); dt = (time.time() - t) / 10
        print("plan", plan, "%.2f ms  %.1f images/s" % (dt * 1e3, 16 / dt), flush=True)
        del g
