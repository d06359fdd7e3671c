"""Numerics of Winograd F(2x2,3x3) with f16 operands on the dense-block shapes (no GPU needed): the transformed input
V = B^T d B and the transformed weights U = G g G^T are rounded to f16 (what an MFMA operand holds), products accumulate in
fp32, the output transform runs in fp32 -- against the fp32 convolution, next to the direct f16-operand convolution (what fast
mode computes today).  Input transform evaluated in f16 arithmetic (packed-f16 VALU) or in fp32 before the rounding.

    python tools/winograd_numerics.py
"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd(x, w, transform_dtype):
    n, c, h, wd = x.shape
    co = w.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    # 4x4 patches with stride 2: [n, c, th, tw, 4, 4]
    p = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    d = p.to(transform_dtype)
    bt = BT.to(transform_dtype)
    v = torch.einsum("ij,nctujk,lk->nctuil", bt, d, bt)          # B^T d B in the transform's arithmetic
    v = v.to(torch.float16).float()                                # MFMA operand
    u = torch.einsum("ij,ocjk,lk->ocil", G, w, G).to(torch.float16).float()
    m = torch.einsum("ocil,nctuil->notuil", u, v)                  # fp32 accumulate over channels, per position
    y = torch.einsum("ij,notujk,lk->notuil", AT, m, AT)            # [n, co, th, tw, 2, 2]
    th, tw = y.shape[2], y.shape[3]
    return y.permute(0, 1, 2, 4, 3, 5).reshape(n, co, th * 2, tw * 2)[:, :, :h, :wd]


def main():
    rows = []
    for cin, cout in ((64, 32), (160, 32), (192, 64)):
        # activations like a dense block's: post-LeakyReLU features of O(1), weights kaiming * 0.1 (model.py:100-106)
        x = F.leaky_relu(torch.randn(2, cin, 32, 32), 0.2)
        x = x.to(torch.float16).float()                            # stored f16 activations
        w = torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5 * 0.1
        ref = F.conv2d(x.double(), w.double(), padding=1).float()
        direct = F.conv2d(x, w.to(torch.float16).float(), padding=1)
        wino16 = winograd(x, w, torch.float16)
        wino32 = winograd(x, w, torch.float32)
        scale = ref.abs().max().item()
        rows.append((cin, cout, scale, (direct - ref).abs().max().item(), (wino32 - ref).abs().max().item(), (wino16 - ref).abs().max().item()))
    print(f"{'cin->cout':>10s} {'max|y|':>8s} {'direct f16 operands':>20s} {'winograd, fp32 input transform':>32s} {'winograd, f16 input transform':>30s}")
    for cin, cout, sc, a, b, c in rows:
        print(f"{cin:4d}->{cout:<4d} {sc:8.3f} {a:20.2e} {b:32.2e} {c:30.2e}")


if __name__ == "__main__":
    main()
