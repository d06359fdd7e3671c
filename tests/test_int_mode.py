"""Integer mode of blur / resize / JPEG (north_star: "blur/resize/JPEG bit-exact in integer mode").
CPU part: the numpy oracle (oracle/imgproc_int_ref.py) against the reference's own float outputs on the same uint8 images
(tests/golden/imgproc_filter.npz, imgproc_resize.npz, imgproc_jpeg.npz: <= 1 LSB; JPEG's quantised coefficients equal to the
reference's except on rounding ties), and the product's host-side quantisers against the oracle's.
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


JPEG_TAGS = ("48x40", "77x77", "100x100")       # not multiples of 16: the zero padding is part of the round trip


def test_oracle_jpeg_against_the_reference_float_round_trip():
    """The integer round trip on the goldens' uint8 images (qualities 30, 49.9, 50, 95): quantised coefficients equal to the
    ones the reference's own compress_jpeg produced (a float-rounding tie may flip one: < 1e-4 of them), output <= 1 LSB from
    rint(255 * DiffJPEG(x))."""
    z = np.load(os.path.join(G, "imgproc_jpeg.npz"))
    total = bad = 0
    for tag in JPEG_TAGS:
        xu = u8(z["x_" + tag])
        out, coefs = J.jpeg_u8(xu, z["q_" + tag], return_coefficients=True)
        for got, key in zip(coefs, ("cy_", "ccb_", "ccr_")):
            ref = z[key + tag]
            bad += int((got.reshape(ref.shape) != ref).sum())
            total += ref.size
        assert np.abs(out.astype(int) - np.rint(z["y_" + tag] * 255).astype(int)).max() <= 1, tag
    assert bad <= 1e-4 * total, (bad, total)
    C = J.jpeg_dct_matrix()
    assert np.abs(C @ C.T - np.eye(8) * (1 << 40)).max() < 1 << 24        # orthonormal up to the rounding of its entries (2^-16 relative)
    assert J._TO_YCC.sum(axis=1).tolist() == [1 << 20, 0, 0]               # grey stays grey
    grey = np.full((1, 3, 16, 16), 77, dtype=np.uint8)
    assert np.array_equal(J.jpeg_u8(grey, np.float32([90.0])), grey)        # a constant block survives exactly


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


@pytest.mark.gpu
def test_hip_integer_jpeg_equals_the_oracle_bit_for_bit():
    from real_esrgan_pytorch_amd import imgproc
    z = np.load(os.path.join(G, "imgproc_jpeg.npz"))
    rng = np.random.default_rng(5)
    cases = [(u8(z["x_" + tag]), z["q_" + tag]) for tag in JPEG_TAGS]
    cases.append((rng.integers(0, 256, size=(3, 3, 5, 7), dtype=np.uint8), np.float32([10.0, 50.0, 99.5])))       # smaller than one block
    cases.append((rng.integers(0, 256, size=(2, 3, 64, 48), dtype=np.uint8), np.float32([1.0, 100.0])))           # extreme steps; 100 -> step 1 (clamped from 0)
    smooth = np.clip(np.add.outer(np.arange(90), np.arange(130)) * 1.1 + rng.normal(0, 2, (2, 3, 90, 130)), 0, 255).astype(np.uint8)
    cases.append((smooth, np.float32([35.5, 72.25])))                                                           # image-like: mostly zero coefficients
    sat = np.zeros((1, 3, 32, 32), dtype=np.uint8)
    sat[:, :, ::2] = 255
    cases.append((sat, np.float32([30.0])))                                                                     # overshoot on both clamps
    worst = 0
    for xu, q in cases:
        want, wc = J.jpeg_u8(xu, q, return_coefficients=True)
        qt = torch.from_numpy(q.copy()).cuda()
        got, gc = imgproc.jpeg_u8(torch.from_numpy(xu).cuda(), qt, return_coeffs=True)
        assert np.array_equal(got.cpu().numpy(), want), (xu.shape, np.abs(got.cpu().numpy().astype(int) - want.astype(int)).max())
        flat = np.concatenate([c.reshape(c.shape[0], -1, 64) for c in wc], axis=1)
        assert np.array_equal(gc.cpu().numpy(), flat), xu.shape
        f64 = q.astype(np.float64)
        assert np.allclose(qt.cpu().numpy(), np.where(f64 < 50, 50 / f64, 2 - f64 / 50), rtol=1e-6)                # the reference's in-place quality -> factor quirk
    for tag in JPEG_TAGS:                                # distance to the reference's float DiffJPEG, in LSBs
        got = imgproc.jpeg_u8(torch.from_numpy(u8(z["x_" + tag])).cuda(), torch.from_numpy(z["q_" + tag].copy()).cuda()).cpu().numpy()
        worst = max(worst, int(np.abs(got.astype(int) - np.rint(z["y_" + tag] * 255).astype(int)).max()))
    assert worst <= 1
    same = imgproc.jpeg_u8(torch.from_numpy(cases[0][0]).cuda(), 70)                                             # scalar quality
    assert np.array_equal(same.cpu().numpy(), J.jpeg_u8(cases[0][0], np.full(4, 70, dtype=np.float32)))
    with pytest.raises(ValueError):
        imgproc.jpeg_u8(torch.zeros(1, 1, 16, 16, dtype=torch.uint8, device="cuda"), 50)
