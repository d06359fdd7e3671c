"""USMSharp(50, 0).forward alone: the two fused launches (csrc/degrade.hip usm51_kernel) against the six separate passes
(RESR_USM_SIX_PASSES=1), the degradation's HR batch (16 x 3 x 1024^2), the GAN step's sr batch (16 x 3 x 256^2) and the
reference's 400^2 tiles; event-timed, device otherwise idle.

    python tools/time_usm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from real_esrgan_pytorch_amd import imgproc  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


usm = imgproc.USMSharp(50, 0).cuda()
for shape in ((16, 3, 1024, 1024), (16, 3, 400, 400), (16, 3, 256, 256), (32, 3, 256, 256)):
    x = torch.rand(*shape, device="cuda")
    px = x.numel()
    row = {}
    for six in (True, False):
        if six:
            os.environ["RESR_USM_SIX_PASSES"] = "1"
        else:
            os.environ.pop("RESR_USM_SIX_PASSES", None)
        with torch.no_grad():
            ms = timed(lambda: usm(x, 0.5, 10))
        row["six passes" if six else "two launches"] = ms
    os.environ.pop("RESR_USM_SIX_PASSES", None)
    print(f"{shape}: six passes {row['six passes']:.3f} ms ({60.0 * px / row['six passes'] / 1e6:.0f} GB/s of its 60 B/value), "
          f"two launches {row['two launches']:.3f} ms ({22.0 * px / row['two launches'] / 1e6:.0f} GB/s of its 22 B/value)  x{row['six passes'] / row['two launches']:.2f}", flush=True)
