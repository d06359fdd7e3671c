"""CPU: oracle/imgproc_ref.py against golden vectors produced by the reference's imgproc.py."""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from oracle import imgproc_ref as I

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k] for k in z.files}


def err(a, b):
    return (a - b).abs().max().item()


def test_usm_and_filter2d():
    g = load("imgproc_filter")
    assert torch.equal(I.usm_kernel(50, 0), g["usm_kernel"])
    assert err(I.usm_sharp(g["x"], g["usm_kernel"], 0.5, 10), g["usm"]) < 1e-6
    assert err(I.filter2d(g["x"], g["k7"]), g["f7"]) < 1e-6
    assert err(I.filter2d(g["x"], g["k21"]), g["f21"]) < 1e-6


def test_resize_call_sites():
    g = load("imgproc_resize")
    for mode in ("area", "bilinear", "bicubic"):
        for s in (0.3731, 1.3177):
            assert torch.equal(F.interpolate(g["x"], scale_factor=s, mode=mode), g[f"{mode}_sf_{s}"])
        assert torch.equal(F.interpolate(g["x"], size=(30, 27), mode=mode), g[f"{mode}_size_30x27"])


def test_noise_same_seed_same_draw_order():
    g = load("imgproc_noise")
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        assert err(I.random_add_gaussian_noise(g["x"], [1, 30], 0.4), g[f"gauss_{seed}"]) < 1e-6, seed
        torch.manual_seed(seed)
        assert err(I.random_add_poisson_noise(g["x"], [0.05, 3], 0.4), g[f"poisson_{seed}"]) < 1e-6, seed
    torch.manual_seed(4)
    assert err(I.random_add_gaussian_noise(g["x"], [1, 25], 0.0), g["gauss_nogray"]) < 1e-6
    torch.manual_seed(4)
    assert err(I.random_add_poisson_noise(g["x"], [0.05, 2.5], 1.0), g["poisson_allgray"]) < 1e-6


def test_diff_jpeg():
    g = load("imgproc_jpeg")
    for tag in ("48x40", "77x77", "100x100"):
        y, c = I.diff_jpeg(g[f"x_{tag}"], g[f"q_{tag}"], return_coeffs=True)
        assert torch.allclose(I.quality_to_factor(g[f"q_{tag}"]), g[f"factor_{tag}"], rtol=1e-6, atol=0)
        for name, key in (("y", "cy"), ("cb", "ccb"), ("cr", "ccr")):
            assert torch.equal(c[name], g[f"{key}_{tag}"]), (tag, name)      # integer coefficients: exact
        assert err(y, g[f"y_{tag}"]) < 1e-5, tag
    assert torch.allclose(I.quality_to_factor(torch.tensor([30.0, 49.9, 50.0, 95.0])), g["q_mutated"], rtol=1e-6, atol=0)
    assert err(I.diff_jpeg(g["x_48x40"], torch.full((4,), 70.0)), g["y_scalar_q70"]) < 1e-5


def test_quantize_and_crop():
    g = load("imgproc_crop")
    plr, phr = I.crop_pair(g["lr"], g["hr"], 64, 4, int(g["top"]), int(g["left"]))
    assert torch.equal(plr, g["plr"]) and torch.equal(phr, g["phr"])
    x = torch.tensor([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 1.2, -0.3])
    assert torch.equal(I.quantize(x) * 255, torch.tensor([0.0, 0.0, 2.0, 2.0, 255.0, 0.0]))     # half-to-even


def test_kernel_synthesis():
    g = load("imgproc_kernels")
    chk = lambda a, b: np.allclose(a, b.numpy(), rtol=1e-12, atol=1e-15)
    for iso, t in ((True, "iso"), (False, "aniso")):
        assert chk(I.bivariate_kernel("gaussian", 21, 2.1, 0.9, 0.7, isotropic=iso), g[f"gauss_{t}"])
        assert chk(I.bivariate_kernel("generalized", 15, 1.7, 0.6, -1.1, 2.3, iso), g[f"general_{t}"])
        assert chk(I.bivariate_kernel("plateau", 9, 2.6, 1.2, 2.0, 1.4, iso), g[f"plateau_{t}"])
    assert chk(I.sinc_kernel(2.5, 7, 0), g["sinc_7"]) and chk(I.sinc_kernel(1.1, 13, 21), g["sinc_13_pad21"])
    for seed in range(8):
        random.seed(seed)
        np.random.seed(seed)
        k = I.random_mixed_kernel([0.45, 0.25, 0.12, 0.03, 0.12, 0.03], 7 + 2 * seed, [0.2, 3], [-np.pi, np.pi], [0.5, 4], [1, 2])
        assert chk(k, g[f"mixed_{seed}"]), seed
