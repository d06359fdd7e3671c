"""CPU tests: the oracle's restatement of the WHOLE degradation loop body + one RealESRNet step (oracle/degrade_ref.py)
against the reference's own `train()` executed for one batch (tests/golden/pipeline_seed*.npz, made by
tests/golden/gen_pipeline_golden.py), and the product's host-side draw order (degrade.sample_plan) against the decisions the
reference took under the same seeds (reference train_realesrnet.py:262-397)."""
import glob
import os
import random

import numpy as np
import pytest
import torch

from oracle import degrade_ref as D
from oracle import model_ref as M

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "pipeline_seed*.npz")))
STAGES = ["usm", "blur1", "resize1", "noise1", "jpeg1", "blur2", "resize2", "noise2", "resize3", "sinc", "jpeg2", "lr_full"]


def load_case(path):
    z = np.load(path)
    plan = {k[5:]: z[k].item() for k in z.files if k.startswith("plan_")}
    draws = [(k.split("_", 2)[2], torch.from_numpy(z[k])) for k in sorted(z.files) if k.startswith("draw_")]
    t = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("t_")}
    return z, plan, draws, t


def process_params():
    from real_esrgan_pytorch_amd import config
    return config.degradation_process_parameters_dict


def test_golden_inventory():
    assert len(CASES) >= 2
    plans = [load_case(p)[1] for p in CASES]
    assert {p["noise1_gaussian"] for p in plans} == {True, False}      # both noise kinds in both positions
    assert {p["noise2_gaussian"] for p in plans} == {True, False}
    assert {p["sinc_before_jpeg"] for p in plans} == {True, False}     # both orders of the final block


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_pipeline_matches_reference_loop(path):
    z, plan, draws, t = load_case(path)
    P = process_params()
    src = D.Draws(draws)
    trace = {}
    lr, hrc = D.degrade_batch(torch.from_numpy(z["hr"]), torch.from_numpy(z["k1"]), torch.from_numpy(z["k2"]),
                              torch.from_numpy(z["ksinc"]), plan, P, 4, 64, src, trace)
    assert src.exhausted(), "the oracle consumed the device draws in a different order than the reference"
    assert torch.equal(trace["q1"], t["q1"]) and torch.equal(trace["q2"], t["q2"])
    for name in STAGES:
        assert trace[name].shape == t[name].shape, name
        err = (trace[name] - t[name]).abs().max().item()
        assert err < 2e-6, (name, err)
    assert torch.equal(lr, t["lr"]) and torch.equal(hrc, t["hr_crop"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_host_draw_order_matches_reference(path):
    """Seeding `random`/`np.random` like config.py:64-66, both host samplers reproduce the decisions the reference's loop took."""
    from real_esrgan_pytorch_amd import degrade
    z, plan, _, _ = load_case(path)
    seed = int(z["seed"])
    P = process_params()
    hr_size = z["hr"].shape[-1]
    random.seed(seed)
    np.random.seed(seed)
    got = D.sample_plan(hr_size, hr_size, 64, P)
    for k, v in plan.items():
        assert got[k] == v or (isinstance(v, float) and abs(got[k] - v) < 1e-15), (k, got[k], v)
    random.seed(seed)
    np.random.seed(seed)
    prod = degrade.sample_plan(2, hr_size, hr_size, 64, with_kernels=False)      # the product's host logic
    for k, v in plan.items():
        pv = getattr(prod, k)
        assert pv == v or (isinstance(v, float) and abs(pv - v) < 1e-15), (k, pv, v)


@pytest.mark.parametrize("path", CASES[:1], ids=[os.path.basename(p)[:-4] for p in CASES[:1]])
def test_oracle_realesrnet_step_matches_reference(path):
    """L1 loss, SR output and all 702 gradient norms of train_realesrnet.py:379-388 on the golden LR/HR pair."""
    z, _, _, t = load_case(path)
    seed = int(z["seed"])
    sd = M.init_generator_state(40 + seed, 3, 3, 4, bias_noise=0.02)
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, sr = D.realesrnet_step(params, t["lr"], t["hr_crop"])
    assert abs(loss.item() - float(t["loss"])) < 1e-6
    assert (sr - t["sr"]).abs().max().item() < 1e-5
    norms = torch.stack([params[k].grad.norm() for k in sd])       # named_parameters order == state_dict order here
    ref = torch.from_numpy(z["grad_norms"])
    assert norms.shape == ref.shape
    rel = ((norms - ref).abs() / ref.clamp_min(1e-12)).max().item()
    assert rel < 1e-3, rel
    for k in ("conv1.weight", "trunk.11.rdb2.conv3.weight", "conv4.bias"):
        g = torch.from_numpy(z["g_" + k])
        assert ((params[k].grad - g).norm() / g.norm()).item() < 1e-4, k


def test_oracle_gan_step_matches_reference():
    """The oracle's RealESRGAN step (oracle/degrade_ref.py:realesrgan_step) against the reference's own `train()` of
    train_realesrgan.py executed for one batch (tests/golden/gan_step_seed5.npz): the four losses, SR, the gradient norm of
    every generator / discriminator tensor and the spectral-norm vectors after the step's three discriminator calls."""
    z = np.load(os.path.join(HERE, "golden", "gan_step_seed5.npz"))
    seed = int(z["seed"])
    gsd = M.init_generator_state(60 + seed, 3, 3, 4, bias_noise=0.02)
    gsd["conv4.bias"] = gsd["conv4.bias"] + 0.5
    dsd = M.init_discriminator_state(80 + seed)
    gp = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    dp = {k: v.clone() for k, v in dsd.items()}
    for k in dp:
        if not (k.endswith("_u") or k.endswith("_v")):
            dp[k].requires_grad_(True)
    losses, sr = D.realesrgan_step(gp, dp, torch.from_numpy(z["lr"]), torch.from_numpy(z["hr_crop"]))
    for k, v in losses.items():
        assert abs(v.item() - float(z[k])) < 2e-6, (k, v.item(), float(z[k]))
    assert (sr - torch.from_numpy(z["sr"])).abs().max().item() < 1e-5
    gn = torch.stack([gp[k].grad.norm() for k in gsd])
    ref = torch.from_numpy(z["g_grad_norms"])
    assert ((gn - ref).abs() / ref.clamp_min(1e-12)).max().item() < 2e-3
    dkeys = [k for k in dsd if not (k.endswith("_u") or k.endswith("_v"))]
    dn = torch.stack([dp[k].grad.norm() for k in _ref_param_order(dkeys)])
    ref = torch.from_numpy(z["d_grad_norms"])
    assert dn.shape == ref.shape and ((dn - ref).abs() / ref.clamp_min(1e-12)).max().item() < 2e-3
    for k in z.files:
        if k.startswith("uv_"):
            assert torch.allclose(dp[k[3:]], torch.from_numpy(z[k]), atol=2e-6), k
        if k.startswith("dg_"):
            g = torch.from_numpy(z[k])
            assert ((dp[k[3:]].grad - g).norm() / g.norm()).item() < 1e-4, k


def _ref_param_order(keys):
    """named_parameters() order of the reference Discriminator (model.py:136-175): a spectral-norm conv registers its
    `weight_orig` where `weight` stood, biases follow weights."""
    order = ["conv1.weight", "conv1.bias", "down_block1.0.weight_orig", "down_block2.0.weight_orig", "down_block3.0.weight_orig",
             "up_block1.0.weight_orig", "up_block2.0.weight_orig", "up_block3.0.weight_orig", "conv2.0.weight_orig",
             "conv3.0.weight_orig", "conv4.weight", "conv4.bias"]
    assert sorted(order) == sorted(keys), (order, keys)
    return order
