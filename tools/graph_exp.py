import sys, time, torch
sys.path.insert(0, '.')
import real_esrgan_pytorch_amd as R
torch.manual_seed(0)
g = R.Generator(3, 3, 4, precision="fast").cuda().eval()
for B, S in ((32, 64), (16, 64), (16, 256)):
    x = torch.rand(B, 3, S, S, device="cuda")
    with torch.no_grad():
        for _ in range(3): y = g(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): y = g(x)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            y2 = g(x)
        gr.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
    print(f"B={B} S={S}: eager {te*1e3:.3f} ms, graph {tg*1e3:.3f} ms, ratio {te/tg:.3f}")
