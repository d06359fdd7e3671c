"""Does a second, independent half batch on another stream fill the gaps between DEPENDENT kernels of a small-image pass?
Generator forward (inference) at LR 64^2: one graph of the whole batch vs two graphs of half the batch each (two model
instances with the same weights, so two workspaces), launched on two streams."""
import sys, time, torch
sys.path.insert(0, '.')
import real_esrgan_pytorch_amd as R
torch.manual_seed(0)
g1 = R.Generator(3, 3, 4, precision="fast").cuda().eval()
g2 = R.Generator(3, 3, 4, precision="fast").cuda().eval()
g2.load_state_dict(g1.state_dict())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, S in ((32, 64), (16, 64), (64, 64), (8, 128)):
    x = torch.rand(B, 3, S, S, device="cuda")
    xa, xb = x[:B // 2].contiguous(), x[B // 2:].contiguous()
    with torch.no_grad():
        for _ in range(2):
            g1(x); g1(xa); g2(xb)
        torch.cuda.synchronize()
        full = torch.cuda.CUDAGraph()
        with torch.cuda.graph(full):
            y = g1(x)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            ya = g1(xa)
        with torch.cuda.graph(gb):
            yb = g2(xb)
        t_full = timed(full.replay)

        def both():
            with torch.cuda.stream(s1):
                ga.replay()
            with torch.cuda.stream(s2):
                gb.replay()
        t_two = timed(both)

        def serial():
            ga.replay(); gb.replay()
        t_ser = timed(serial)
    print(f"B={B} S={S}: one graph {t_full:.3f} ms | two half-batch graphs on two streams {t_two:.3f} ms | the same two in series {t_ser:.3f} ms")
