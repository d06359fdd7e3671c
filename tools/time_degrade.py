import sys, time, torch
sys.path.insert(0, "/root/repo")
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from real_esrgan_pytorch_amd.degrade import Degrader
B = 16
d = Degrader(batch=B, hr_size=1024, upscale=4, crop=1024, seed=0)
hr = torch.rand(B, 3, 1024, 1024, device="cuda")
for _ in range(3): lr, h = d(hr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): lr, h = d(hr)
torch.cuda.synchronize(); print("degrade ms/step", (time.perf_counter() - t0) / 10 * 1e3)
