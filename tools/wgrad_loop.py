"""One weight-gradient launch (cin 192 -> cout 64: twelve products, three quad jobs x splits) repeated back to back for ~1.5 s at
several batch sizes: does the kernel itself run faster on more pixels per workgroup, or only inside the step?"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib
lib = L.lib()
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--batches", default="8,16,32")
ap.add_argument("--splits", type=int, default=0)
a = ap.parse_args()
for n in map(int, a.batches.split(",")):
    cin, cout, h, w = 192, 64, a.res, a.res
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand(n, h, w, cin, device="cuda", generator=gen) - 0.5).half()      # interleaved NHWC (the public descriptor)
    g = (torch.rand(n, h, w, cout, device="cuda", generator=gen) - 0.5).half()
    splits = a.splits or 84   # 3 quad jobs x 84 = 252 workgroups
    d = L.WgradDesc(n, h, w, cin, cin, cin, 0, cin, cout, cout, cout, L.RESR_F16, 0, splits, 1.0)
    nbytes = lib.resr_wgrad_partial_bytes(C.byref(d))
    partial = torch.empty(nbytes // 4, device="cuda")
    dw = torch.empty(cout, cin, 3, 3, device="cuda"); db = torch.empty(cout, device="cuda")
    def launch():
        L.check(lib.resr_conv3x3_wgrad(C.byref(d), L.ptr(x), None, L.ptr(g), L.ptr(partial), L.ptr(dw), L.ptr(db), L.stream_ptr()))
    for _ in range(3): launch()
    torch.cuda.synchronize()
    reps = max(10, int(1.5 / (2.0 * 9 * cin * cout * n * h * w / 1.1e15)))
    t0 = time.perf_counter()
    for _ in range(reps): launch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:3d} {h}x{w}: {dt * 1e6:8.1f} us per launch pair (wgrad + reduce)  {2.0 * 9 * cin * cout * n * h * w / dt / 1e12:7.1f} TFLOP/s")
