"""GPU diagnostic: compare backward intermediates of the tail with the oracle (strict mode)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import real_esrgan_pytorch_amd as R
from oracle import model_ref as M
L = R._lib
NAMES = ["x_in", "ws0", "out1", "trunk_out", "feat", "u1", "u2", "c3", "ymask", "g4", "gA", "gB", "gM1", "gF",
         "gT0", "gT1", "gT2", "gT3", "gS", "gxin", "partial"]

def run(n, h, w, nb=1, up=4):
    sd = M.init_generator_state(11, 3, 3, up, bias_noise=0.02)
    sd = {k: v for k, v in sd.items() if not k.startswith("trunk.") or int(k.split(".")[1]) < nb}
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    g = R.Generator(3, 3, up, precision="strict", n_blocks=nb); g.load_state_dict(sd); g = g.cuda()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, h, w, generator=gen); gw = torch.randn(n, 3, h*up, w*up, generator=gen)
    # oracle with retained intermediates
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    def conv(t, k): return F.conv2d(t, sdo[k + ".weight"], sdo[k + ".bias"], padding=1)
    out1 = conv(x, "conv1"); t = out1
    for i in range(nb): t = M.rrdb_forward(t, sdo, f"trunk.{i}")
    trunk_out = t; trunk_out.retain_grad()
    feat = out1 + conv(trunk_out, "conv2"); feat.retain_grad()
    p1 = conv(F.interpolate(feat, scale_factor=2, mode="nearest"), "upsampling1.0"); p1.retain_grad()
    u1 = F.leaky_relu(p1, 0.2)
    p2 = conv(F.interpolate(u1, scale_factor=2, mode="nearest"), "upsampling2.0"); p2.retain_grad()
    u2 = F.leaky_relu(p2, 0.2)
    p3 = conv(u2, "conv3.0"); p3.retain_grad()
    c3 = F.leaky_relu(p3, 0.2)
    y = conv(c3, "conv4").clamp(0, 1)
    (y * gw).sum().backward()
    xd = x.cuda().requires_grad_(True)
    yd = g(xd)
    (yd * gw.cuda()).sum().backward()
    torch.cuda.synchronize()
    desc = g._desc(xd.detach(), True)
    offs = (C.c_int64 * 32)()
    cnt = L.lib().resr_generator_buffer_offsets(C.byref(desc), C.cast(offs, C.c_void_p), 32)
    off = dict(zip(NAMES, list(offs)[:cnt]))
    ws = next(iter(g._workspaces.values()))[0].buf
    def view(name, hh, ww, c):
        o = off[name]
        return ws[o:o + n*hh*ww*c*4].view(torch.float32).view(n, hh, ww, c).permute(0, 3, 1, 2).cpu()
    def rep(name, got, ref):
        e = (got - ref).abs()
        per = [e[i].max().item() / ref.abs().max().item() for i in range(n)]
        print(f"   {name:10s} rel err per image {['%.2e' % v for v in per]}")
    print(f"n={n} {h}x{w}")
    rep("fwd y", yd.detach().cpu(), y.detach())
    rep("gA=G_conv3", view("gA", 4*h, 4*w, 64), p3.grad)
    rep("u2 (fwd)", view("u2", 4*h, 4*w, 64), u2.detach())
    rep("gB=G_up2", view("gB", 4*h, 4*w, 64), p2.grad)
    eb = (view("gB", 4*h, 4*w, 64) - p2.grad).abs().amax(dim=1)   # [n,H,W]
    bad = (eb > 1e-4 * p2.grad.abs().max()).nonzero()
    print("   gB bad pixels:", bad.shape[0], "first", bad[:12].tolist(), "last", bad[-6:].tolist())
    rep("gM1=G_up1", view("gM1", 2*h, 2*w, 64), p1.grad)
    rep("gF=g_feat", view("gF", h, w, 64), feat.grad)
    rep("c3 (fwd)", view("c3", 4*h, 4*w, 64), c3.detach())
    rep("u1 (fwd)", view("u1", 2*h, 2*w, 64), u1.detach())
    for k in ["conv4", "conv3.0", "upsampling2.0", "upsampling1.0", "conv2"]:
        pg = dict(g.named_parameters())[k + ".weight"].grad.cpu()
        print(f"   dW {k:14s} rel err {(pg - sdo[k + '.weight'].grad).abs().max().item() / sdo[k + '.weight'].grad.abs().max().item():.2e}")

for shp in [(2, 32, 32), (2, 33, 17), (2, 20, 24)]:
    run(*shp)
