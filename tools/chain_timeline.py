"""Needs a TRACE BUILD of the library (the hooks are compiled out of the product build):
    python tools/build_variant.py trace -DRESR_TRACE=1 && RESR_LIB_PATH=$PWD/tools/ab/trace.so python tools/chain_timeline.py ...

Timeline of the LAST chained dense-block launch of a generator forward (RESR_TRACE_CHAIN_ONLY=1 keeps the trace buffer for
chained launches): producer / consumer stamps of a few workgroups, in us."""
import argparse, os, sys
os.environ["RESR_TRACE_CHAIN_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R
L = R._lib
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--wgs", default="0,1,8")
a = ap.parse_args()
lib = L.lib()
g = R.Generator(3, 3, 4, precision="fast", n_blocks=1).cuda().train()
x = torch.rand(a.batch, 3, a.res, a.res, device="cuda")
for _ in range(2): g(x)
torch.cuda.synchronize()
tr = torch.zeros(32 * 2 * 64, dtype=torch.int64, device="cuda")
lib.resr_debug_conv_trace(L.ptr(tr))
g(x)
torch.cuda.synchronize()
lib.resr_debug_conv_trace(None)
t = tr.cpu().view(32, 2, 64)
t0 = int(t[t > 0].min())
for wg in map(int, a.wgs.split(",")):
    for role, name in ((0, "producer"), (1, "consumer0")):
        v = [int(q) for q in t[wg, role] if q > 0]
        print(f"wg{wg} {name}: " + " ".join(f"{(q - t0) / 100:.2f}" for q in v))
# producer: six stamps per stage -- after the barrier | weights requested | next stage worked out | polled | halo requested | older requests landed
for wg in map(int, a.wgs.split(",")):
    pr = [int(q) for q in t[wg, 0] if q > 0]
    print(f"wg{wg} producer stages (us): W issue, advance (+ tile index math), poll, halo issue, wait, to next barrier")
    for i in range(1, len(pr) - 6, 6):
        d = [(pr[i + k + 1] - pr[i + k]) / 100 for k in range(6)]
        print(f"   t={(pr[i] - t0) / 100:7.2f}  " + "  ".join(f"{x:5.2f}" for x in d) + f"   period {(pr[i + 6] - pr[i]) / 100:5.2f}")
print("chain errors:", int(lib.resr_debug_chain_errors()))
