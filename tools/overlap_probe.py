"""Does the RealESRGAN step's generator part leave room for the discriminator / VGG19 work that does not depend on it?
Times (a) generator forward + backward at config 4's geometry (16 x 64^2 LR, chained dense blocks at one tile per CU),
(b) D(hr) forward + backward and VGG19(hr) features alone, (c) both on two streams at once.  (c) well under (a) + (b) would
make a second stream for the independent half of the step worth building; (c) ~ (a) + (b) says the launches already own the chip."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import real_esrgan_pytorch_amd as R

torch.manual_seed(0)
B = int(os.environ.get("B", 16))
g = R.Generator(3, 3, 4).cuda().train()
d = R.Discriminator().cuda().train()
cl = R.ContentLoss(["features.2", "features.7", "features.16", "features.25", "features.34"], [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]).cuda()
lr = torch.rand(B, 3, 64, 64, device="cuda")
hr = torch.rand(B, 3, 256, 256, device="cuda")
side = torch.cuda.Stream()


def gen():
    g.zero_grad(set_to_none=False)
    y = g(lr)
    y.sum().backward()


def disc():
    d.zero_grad(set_to_none=False)
    o = d(hr)
    o.sum().backward()
    with torch.no_grad():
        cl(hr, hr)


enq = {}


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    enq[fn.__name__] = round((t1 - t0) / n * 1e3, 3)     # host time to enqueue one call: ~ the total when the host is the bound
    return (time.perf_counter() - t0) / n * 1e3


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        disc()
    gen()
    torch.cuda.current_stream().wait_stream(side)


def serial():
    disc(); gen()


res = {"gen_ms": timed(gen), "disc_vgg_ms": timed(disc), "serial_ms": timed(serial), "two_streams_ms": timed(both),
       "chain_errors": int(R._lib.lib().resr_debug_chain_errors())}
res = {k: round(v, 3) for k, v in res.items()}
res["host_enqueue_ms"] = enq
print(json.dumps(res))
