"""Inference configurations of BASELINE.json (parity-test cases, not the headline bench):
  config 2: RRDBNet x4 f16, batch 16 of 256x256 LR;  config 5: RRDBNet x2, 3840x2160 LR, tiled + hipGraph."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
from real_esrgan_pytorch_amd.tiling import TiledGenerator

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

torch.manual_seed(0)
g4 = R.Generator(3, 3, 4, precision="fast").cuda().eval()
x = torch.rand(16, 3, 256, 256, device="cuda")
with torch.no_grad():
    dt = timeit(lambda: g4(x), 5)
flop = 2 * 17_926_848 * 256 * 256 * 16
print(json.dumps({"config": "2: x4 f16 inference, batch 16 of 256^2", "ms": round(dt * 1e3, 2), "images_per_s": round(16 / dt, 1),
                  "tflops": round(flop / dt / 1e12, 1)}))
del g4
g2 = R.Generator(3, 3, 2, precision="fast").cuda().eval()
frame = torch.rand(1, 3, 2160, 3840, device="cuda")
tile = os.environ.get("TILE")
tg = TiledGenerator(g2, tile=int(tile) if tile else None, halo=32, use_graph=True)
dt = timeit(lambda: tg(frame), 2)
flop = 2 * 17_932_032 * 1920 * 1080
print(json.dumps({"config": "5: x2 f16, 3840x2160 LR tiled (%s, halo 32) + whole-frame hipGraph" % (tg.plan(1, 2160, 3840)[1:],), "ms": round(dt * 1e3, 1),
                  "frames_per_s": round(1 / dt, 3), "tflops": round(flop / dt / 1e12, 1)}))
