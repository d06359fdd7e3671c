import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib.lib()
res = {}
for (n, h, w, c) in [(16, 32, 32, 512), (16, 64, 64, 256), (16, 128, 128, 128), (3, 7, 5, 64), (2, 1, 9, 8), (1, 6, 1, 16)]:
    torch.manual_seed(h * 1000 + w)
    src = torch.randn(n, h, w, c, device="cuda").half()
    dst = torch.empty(n, 2 * h, 2 * w, c, device="cuda", dtype=torch.half)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = L.resr_bilinear_up2x(src.data_ptr(), dst.data_ptr(), n, h, w, c, 0, 0, st)
        assert rc == 0, rc
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    ref = torch.nn.functional.interpolate(src.permute(0, 3, 1, 2).float(), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    err = (dst.float() - ref).abs().max().item()
    res[f"{n}x{h}x{w}x{c}"] = {"us": round(us, 1), "TBps": round((src.numel() + dst.numel()) * 2 / us / 1e6, 2), "sha": hashlib.sha1(dst.cpu().numpy().tobytes()).hexdigest()[:12], "err_vs_torch": err}
    gsrc = torch.randn(n, 2 * h, 2 * w, c, device="cuda").half()
    gin = torch.empty(n, h, w, c, device="cuda", dtype=torch.half)
    def runb():
        rc = L.resr_bilinear_up2x(gsrc.data_ptr(), gin.data_ptr(), n, h, w, c, 0, 1, st)
        assert rc == 0, rc
    for _ in range(5): runb()
    e0.record()
    for _ in range(50): runb()
    e1.record(); torch.cuda.synchronize()
    usb = e0.elapsed_time(e1) / 50 * 1e3
    xs = src.permute(0, 3, 1, 2).float().requires_grad_(True)
    torch.nn.functional.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=False).backward(gsrc.permute(0, 3, 1, 2).float())
    errb = (gin.float() - xs.grad.permute(0, 2, 3, 1)).abs().max().item()
    res[f"{n}x{h}x{w}x{c}"].update({"bwd_us": round(usb, 1), "bwd_TBps": round((gsrc.numel() + gin.numel()) * 2 / usb / 1e6, 2),
                                    "bwd_sha": hashlib.sha1(gin.cpu().numpy().tobytes()).hexdigest()[:12], "bwd_err_vs_torch": errb})
print(json.dumps(res))
