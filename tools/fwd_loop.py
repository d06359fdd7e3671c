"""Generator forward only (no optimizer: a timing build with wrong results cannot feed itself garbage), event-timed, with the
output's statistics and the board power next to it.  python tools/fwd_loop.py --batch 32 --res 64 [--train]"""
import argparse, os, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--train", action="store_true", help="train-mode forward (sign words written, chained 4 + closing convolution)")
a = ap.parse_args()
torch.manual_seed(0)
g = R.Generator(3, 3, 4, precision="fast").cuda()
g.train() if a.train else g.eval()
x = torch.rand(a.batch, 3, a.res, a.res, device="cuda")
pw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
def power():
    try:
        return max(int(open(p).read()) for p in pw) / 1e6
    except Exception:
        return float("nan")
with torch.no_grad():
    for _ in range(10): y = g(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ws = []
    for i in range(a.reps):
        y = g(x)
        if i % 8 == 7: ws.append(power())
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
print(f"{a.batch} x {a.res}^2 {'train' if a.train else 'eval'} forward: {ms:.3f} ms  ({a.batch / ms * 1e3:.1f} images/s)  out mean {float(y.float().mean()):.4f} std {float(y.float().std()):.4f} "
      f"finite {bool(torch.isfinite(y).all())}  power ~{sum(ws) / max(1, len(ws)):.0f} W  chain errors {R._lib.chain_health(sync=True) if hasattr(R._lib, 'chain_health') else '-'}")
