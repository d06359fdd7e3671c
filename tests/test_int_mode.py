"""Integer mode of blur / resize (north_star: "blur/resize/JPEG bit-exact in integer mode").
CPU part: the numpy oracle (oracle/imgproc_int_ref.py) against the reference's own float outputs on the same uint8 images
(tests/golden/imgproc_filter.npz, imgproc_resize.npz: <= 1 LSB), and the product's host-side quantisers against the oracle's.
GPU part (-m gpu): the HIP kernels through the C-ABI, bit for bit against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import imgproc_int_ref as J

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def u8(x):
    return np.rint(np.clip(x, 0, 1) * 255).astype(np.uint8)


def resize_cases():
    z = np.load(os.path.join(G, "imgproc_resize.npz"))
    for k in z.files:
        if k == "x":
            continue
        mode = k.split("_")[0]
        scale = float(k.split("_")[2]) if "_sf_" in k else None
        yield k, mode, scale, z["x"], z[k]


def test_oracle_within_one_lsb_of_the_reference_float_ops():
    z = np.load(os.path.join(G, "imgproc_filter.npz"))
    xu = u8(z["x"])
    for kn, fn in (("k7", "f7"), ("k21", "f21")):
        q = J.quantize_kernel(z[kn])
        assert (q.reshape(q.shape[0], -1).sum(axis=1) == 1 << 14).all()          # unit DC gain survives quantisation
        out = J.filter2d_u8(xu, q)
        assert np.abs(out.astype(int) - u8(z[fn]).astype(int)).max() <= 1, kn
    for name, mode, scale, x, ref in resize_cases():
        out = J.resize_u8(u8(x), ref.shape[2:], scale, mode)
        assert out.shape == ref.shape and np.abs(out.astype(int) - u8(ref).astype(int)).max() <= 1, name


def test_product_host_quantisers_match_the_oracle():
    from real_esrgan_pytorch_amd import imgproc
    rng = np.random.default_rng(3)
    k = rng.normal(size=(3, 21, 21))
    k /= k.sum(axis=(1, 2), keepdims=True)                  # signed taps (sinc-like), unit sum
    assert np.array_equal(imgproc.quantize_kernel_q14(k), J.quantize_kernel(k))
    assert np.array_equal(imgproc.quantize_kernel_q14(torch.from_numpy(k).float()), J.quantize_kernel(k.astype(np.float32)))
    for mode in ("bilinear", "bicubic"):
        for n_in, n_out, scale in ((48, 17, 0.3731), (40, 52, 1.3177), (48, 30, None), (40, 10, None), (7, 7, None), (1, 5, None)):
            if scale is not None:
                n_out = int(np.floor(n_in * scale))
            idx, w = imgproc.resize_tap_tables(n_in, n_out, scale, mode)
            ridx, rw = J.axis_tables(n_in, n_out, scale, mode)
            assert np.array_equal(idx, ridx) and np.array_equal(w, rw), (mode, n_in, n_out, scale)
            assert (w.sum(axis=1) == 1 << 11).all() and idx.min() >= 0 and idx.max() < n_in


@pytest.mark.gpu
def test_hip_integer_blur_and_resize_equal_the_oracle_bit_for_bit():
    from real_esrgan_pytorch_amd import imgproc
    from oracle import imgproc_ref as I
    rng = np.random.default_rng(11)
    z = np.load(os.path.join(G, "imgproc_filter.npz"))
    cases = [(u8(z["x"]), z["k7"]), (u8(z["x"]), z["k21"])]
    sinc = np.stack([I.sinc_kernel(1.3, 21), I.sinc_kernel(2.6, 13, 21)])      # negative taps, zero-padded support
    cases.append((rng.integers(0, 256, size=(2, 3, 37, 53), dtype=np.uint8), sinc))
    cases.append((rng.integers(0, 256, size=(1, 1, 11, 11), dtype=np.uint8), rng.dirichlet(np.ones(441)).reshape(1, 21, 21)))   # kernel wider than the image
    cases.append((rng.integers(0, 256, size=(3, 3, 100, 67), dtype=np.uint8), rng.dirichlet(np.ones(9), size=3).reshape(3, 3, 3)))
    for xu, k in cases:
        q = J.quantize_kernel(k)
        want = J.filter2d_u8(xu, q)
        got = imgproc.filter2d_u8(torch.from_numpy(xu).cuda(), torch.from_numpy(np.asarray(k, dtype=np.float64))).cpu().numpy()
        assert np.array_equal(got, want), (xu.shape, k.shape, np.abs(got.astype(int) - want.astype(int)).max())
    worst = 0
    for name, mode, scale, x, ref in resize_cases():
        xu = u8(x)
        want = J.resize_u8(xu, ref.shape[2:], scale, mode)
        kw = {"scale_factor": scale} if scale is not None else {"size": tuple(ref.shape[2:])}
        got = imgproc.interpolate_u8(torch.from_numpy(xu).cuda(), mode=mode, **kw).cpu().numpy()
        assert np.array_equal(got, want), name
        worst = max(worst, int(np.abs(got.astype(int) - u8(ref).astype(int)).max()))
    assert worst <= 1                                   # distance to the reference's float F.interpolate, in LSBs
    for mode in ("area", "bilinear", "bicubic"):        # ragged sizes, up- and down-scaling, extreme values
        xu = rng.integers(0, 256, size=(2, 3, 61, 45), dtype=np.uint8)
        xu[0, 0, :8] = 255
        xu[1, 2, -8:] = 0
        for size in ((15, 11), (61, 45), (97, 123), (1, 1)):
            want = J.resize_u8(xu, size, None, mode)
            got = imgproc.interpolate_u8(torch.from_numpy(xu).cuda(), size=size, mode=mode).cpu().numpy()
            assert np.array_equal(got, want), (mode, size)
