"""CPU: the oracle (oracle/model_ref.py) against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  fp32 on both sides, same torch kernels -> expect ~1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as M

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-6


def load(name):
    z = np.load(os.path.join(G, name + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) if z[k].dtype.kind == "f" else z[k] for k in z.files}


def close(a, b, tol=TOL):
    scale = max(1.0, b.abs().max().item())
    return (a - b).abs().max().item() <= tol * scale


def test_rdb_forward_backward():
    g = load("rdb")
    sd = {"b." + k[2:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("w_")}
    x = g["x"].clone().requires_grad_(True)
    y = M.rdb_forward(x, sd, "b")
    (y * g["gw"]).sum().backward()
    assert close(y.detach(), g["y"]) and close(x.grad, g["gx"])
    for k, v in sd.items():
        assert close(v.grad, g["g_" + k[2:]], 5e-6), k


def test_rrdb_forward():
    g = load("rrdb")
    sd = {"t." + k[2:]: v.float() for k, v in g.items() if k.startswith("w_")}
    assert close(M.rrdb_forward(g["x"], sd, "t"), g["y"])


@pytest.mark.parametrize("tag", ["x4_a", "x4_b", "x2", "x1"])
def test_generator_forward_backward(tag):
    g = load("generator_" + tag)
    up, seed = int(g["upscale"]), int(g["seed"])
    sd = M.init_generator_state(seed, 3, 3, up, bias_noise=0.02)
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    x = g["x"].clone().requires_grad_(True)
    y = M.generator_forward(x, sd, up)
    assert y.shape == g["y"].shape
    assert close(y.detach(), g["y"])
    (y * g["gw"]).sum().backward()
    assert close(x.grad, g["gx"], 1e-5)
    norms = torch.stack([sd[k].grad.norm() for k in sd])       # reference named_parameters order == key order
    assert torch.allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-7)
    for k in g:
        if k.startswith("g_"):
            assert close(sd[k[2:]].grad, g[k], 1e-5), k


def test_discriminator_three_training_calls_and_eval():
    g = load("discriminator")
    sd = M.init_discriminator_state(int(g["seed"]))
    for k in sd:
        if not (k.endswith("_u") or k.endswith("_v")):
            sd[k].requires_grad_(True)
    x = g["x"].clone().requires_grad_(True)
    for call in range(3):
        for v in sd.values():
            v.grad = None
        x.grad = None
        y = M.discriminator_forward(x, sd, training=True)
        (y * g["gw"]).sum().backward()
        assert close(y.detach(), g[f"y{call}"], 1e-5), call
        assert close(x.grad, g[f"gx{call}"], 1e-5)
        assert close(sd["up_block1.0.weight_u"], g[f"u{call}_up_block1"])
        assert close(sd["down_block3.0.weight_v"], g[f"v{call}_down_block3"])
        assert close(sd["conv1.weight"].grad, g[f"g{call}_conv1.weight"], 1e-5)
        assert close(sd["down_block2.0.weight_orig"].grad[:4], g[f"g{call}_down_block2.weight_orig"], 1e-5)
        assert close(sd["conv3.0.weight_orig"].grad[:8], g[f"g{call}_conv3.weight_orig"], 1e-5)
        assert close(sd["conv4.bias"].grad, g[f"g{call}_conv4.bias"], 1e-5)
    with torch.no_grad():
        assert close(M.discriminator_forward(g["x"], sd, training=False), g["y_eval"], 1e-5)


def _sub(t, cap=4096):      # tests/golden/gen_golden.py _subsample
    f = t.reshape(-1)
    return f if f.numel() <= cap else f[::(f.numel() + cap - 1) // cap]


@pytest.mark.parametrize("name", ["discriminator_allgrads_240", "discriminator_allgrads_243"])
def test_discriminator_every_gradient_tensor(name):
    """Two more seeds without a LeakyReLU pre-activation near zero, EVERY gradient tensor of the three training calls (stored
    subsampled + its norm): the oracle pinned to the reference's own autograd on all of them."""
    g = load(name)
    sd = M.init_discriminator_state(int(g["seed"]))
    names = [str(n) for n in g["names"]]
    for k in names:
        sd[k].requires_grad_(True)
    x = g["x"].clone().requires_grad_(True)
    for call in range(3):
        for v in sd.values():
            v.grad = None
        x.grad = None
        y = M.discriminator_forward(x, sd, training=True)
        (y * g["gw"]).sum().backward()
        assert close(y.detach(), g[f"y{call}"], 1e-5) and close(x.grad, g[f"gx{call}"], 1e-5), call
        for i, k in enumerate(names):
            assert close(_sub(sd[k].grad), g[f"g{call}_{i}"], 1e-5), (call, k)
            assert abs(sd[k].grad.double().norm().item() - float(g[f"n{call}_{i}"])) <= 1e-5 * max(1.0, float(g[f"n{call}_{i}"])), (call, k)


def test_ema_three_updates_bit_exact():
    g = load("ema")
    shadow = {i: g[f"p0_{i}"].clone() for i in range(4)}
    for step in (1, 2, 3):
        M.ema_update(shadow, {i: g[f"p{step}_{i}"] for i in range(4)}, 0.999)
        for i in range(4):
            assert torch.equal(shadow[i], g[f"s{step}_{i}"]), (step, i)
