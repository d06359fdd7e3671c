"""-m gpu: the HIP degradation stage (degrade.run_plan, through the C-ABI) end to end against the reference's own loop body
(train_realesrnet.py:262-377) under the same plan and the reference's recorded device draws, then the RealESRNet step
(train_realesrnet.py:379-388) on the resulting LR / HR pair.  Fixtures: tests/golden/pipeline_seed*.npz
(tests/golden/gen_pipeline_golden.py ran the reference's `train()`)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "pipeline_seed*.npz")))
IDS = [os.path.basename(p)[:-4] for p in CASES]


def load_case(path):
    z = np.load(path)
    plan = {k[5:]: z[k].item() for k in z.files if k.startswith("plan_")}
    draws = [(k.split("_", 2)[2], torch.from_numpy(z[k])) for k in sorted(z.files) if k.startswith("draw_")]
    t = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("t_")}
    return z, plan, draws, t


def build_injection(plan, draws, P, golden):
    """Walk the recorded draws in the reference's call order (imgproc.py:933-936 -> 854/858 or 895/906; train:307,355,361)."""
    it = iter(draws)
    inject, replace = {}, {}

    def nxt(kind):
        k, v = next(it)
        assert k == kind, (k, kind)
        return v
    for tag, rng_key, prng_key, gp_key in (("1", "noise_range1", "poisson_scale_range1", "gray_noise_probability1"),
                                           ("2", "noise_range2", "poisson_scale_range2", "gray_noise_probability2")):
        gaussian = plan[f"noise{tag}_gaussian"]
        raw, rawg = nxt("rand"), nxt("rand")
        gray = (rawg < P[gp_key]).float()
        if gaussian:
            lo, hi = P[rng_key]
            fg = nxt("randn") if gray.sum() > 0 else None
            inject["noise" + tag] = {"sigma": (raw * (hi - lo) + lo).cuda(), "gray": gray.cuda(),
                                     "field_gray": fg.cuda() if fg is not None else None, "field_color": nxt("randn").cuda()}
        else:
            lo, hi = P[prng_key]
            if gray.sum() > 0:
                nxt("poisson")
            nxt("poisson")
            inject["noise" + tag] = {"scale": (raw * (hi - lo) + lo).cuda(), "gray": gray.cuda(), "seed": 7}
            replace["noise" + tag] = golden["noise" + tag].cuda()     # Poisson samples cannot be matched draw for draw
        inject["q" + tag] = nxt("uniform").clone()
    assert next(it, None) is None
    inject["replace"] = replace
    return inject


@pytest.mark.parametrize("path", CASES, ids=IDS)
def test_run_plan_matches_reference_loop(path, diag_dir):
    import json
    from real_esrgan_pytorch_amd import config, degrade, imgproc
    from oracle import imgproc_ref as I
    z, plan, draws, t = load_case(path)
    P = config.degradation_process_parameters_dict
    dp = degrade.DegradationPlan(plan["blur1"], plan["resize1_scale"], plan["resize1_mode"], plan["noise1_gaussian"], plan["blur2"],
                                 plan["resize2_scale"], plan["resize2_mode"], plan["noise2_gaussian"], plan["sinc_before_jpeg"],
                                 plan["resize3_mode"], plan["hr_top"], plan["hr_left"], z["k1"], z["k2"], z["ksinc"])
    inject = build_injection(plan, draws, P, t)
    usm = imgproc.USMSharp(50, 0).cuda()
    jpeg = imgproc.DiffJPEG(False)
    trace = {}
    hr = torch.from_numpy(z["hr"]).cuda()
    lr, hrc = degrade.run_plan(hr, dp, usm, jpeg, 4, 64, trace=trace, inject=inject)
    torch.cuda.synchronize()
    rep = {}
    # Every stage runs on the HIP path's OWN previous output (free run), except that a Poisson stage's output is replaced by
    # the reference's after its input-independent part was checked.  DiffJPEG rounds DCT coefficients: a 1e-6 input
    # difference can flip a coefficient sitting on a .5 tie, which moves one 8x8 block by ~1/255 -- counted, not hidden.
    for name in ("usm", "blur1", "resize1", "noise1", "jpeg1", "blur2", "resize2", "noise2", "resize3", "sinc", "jpeg2"):
        got, ref = trace[name].cpu(), t[name]
        assert got.shape == ref.shape, name
        d = (got - ref).abs()
        rep[name] = {"max": d.max().item(), "frac_gt_2e-5": (d > 2e-5).float().mean().item()}
    with open(os.path.join(diag_dir, f"pipeline_{os.path.basename(path)[:-4]}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    poisson_stages = {f"noise{k}" for k in "12" if not plan[f"noise{k}_gaussian"]}
    seen_jpeg = False
    for name, r in rep.items():
        if name in poisson_stages:
            continue                                   # own Philox samples: checked below by moments
        if not seen_jpeg:
            assert r["max"] < 2e-5, (name, r)
        else:                                          # downstream of a JPEG: allow rare flipped-coefficient blocks
            assert r["frac_gt_2e-5"] < 5e-3 and r["max"] < 0.05, (name, r)
        seen_jpeg = seen_jpeg or name.startswith("jpeg")
    for name in poisson_stages:                        # Poisson stage: E[out - in] ~ 0, same variance scale as the reference's draw
        src = {"noise1": "resize1", "noise2": "resize2"}[name]
        own, ref, base = trace[name].cpu(), t[name], t[src].clamp(0, 1)
        assert abs((own - base).mean().item() - (ref - base).mean().item()) < 6e-3
        assert 0.7 < (own - base).std().item() / max((ref - base).std().item(), 1e-9) < 1.4
    # final LR: quantised to k/255 -- equal to the reference's except where the pre-quantisation value sat within 2e-5 of a .5 tie
    got, ref = lr.cpu(), t["lr"]
    mism = (got != ref)
    assert mism.float().mean().item() < 5e-3, mism.float().mean().item()
    assert ((got - ref).abs()[mism] <= 1.0 / 255 + 1e-6).all() if mism.any() else True
    assert torch.equal(hrc.cpu(), t["hr_crop"])
    assert torch.equal(I.quantize(got), got)


@pytest.mark.parametrize("precision", ["exact16", "strict"])
def test_realesrnet_step_matches_reference(precision):
    """sr = G(lr); loss = L1(sr, hr); backward (train_realesrnet.py:383-388) on the golden LR/HR pair: loss, SR and all 702
    gradient norms against the values the reference's own train() produced."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    z, _, _, t = load_case(CASES[0])
    seed = int(z["seed"])
    sd = M.init_generator_state(40 + seed, 3, 3, 4, bias_noise=0.02)
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    g = R.Generator(3, 3, 4, precision=precision)
    g.load_state_dict(sd)
    g = g.cuda().train()
    scale = 1.0 if precision == "strict" else 4096.0
    sr = g(t["lr"].cuda())
    loss = torch.nn.functional.l1_loss(sr, t["hr_crop"].cuda())
    (loss * scale).backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(t["loss"])) < 2e-6
    assert (sr.detach().cpu() - t["sr"]).abs().max().item() < 1e-4
    norms = torch.stack([p.grad.norm() for p in g.parameters()]).cpu() / scale
    ref = torch.from_numpy(z["grad_norms"])
    rel = ((norms - ref).abs() / ref.clamp_min(1e-12)).max().item()
    assert rel < 2e-3, rel
    for k in ("conv1.weight", "trunk.11.rdb2.conv3.weight", "conv4.bias"):
        gref = torch.from_numpy(z["g_" + k])
        got = dict(g.named_parameters())[k].grad.cpu() / scale
        assert ((got - gref).norm() / gref.norm()).item() < 1e-3, k


@pytest.mark.parametrize("precision", ["strict", "exact16"])
def test_realesrgan_step_matches_reference(precision, diag_dir):
    """train.RealESRGANStep (generator update with the discriminator frozen, USM on sr, three discriminator calls;
    train_realesrgan.py:459-521) on the LR/HR pair of tests/golden/gan_step_seed5.npz against the values the reference's own
    train() produced: the four losses, SR, every gradient norm of both networks and the spectral-norm vectors afterwards."""
    import real_esrgan_pytorch_amd as R
    from oracle import model_ref as M
    from real_esrgan_pytorch_amd.train import RealESRGANStep
    z = np.load(os.path.join(HERE, "golden", "gan_step_seed5.npz"))
    seed = int(z["seed"])
    gsd = M.init_generator_state(60 + seed, 3, 3, 4, bias_noise=0.02)
    gsd["conv4.bias"] = gsd["conv4.bias"] + 0.5
    g = R.Generator(3, 3, 4, precision=precision)
    g.load_state_dict(gsd)
    g = g.cuda().train()
    d = R.Discriminator(precision=precision)         # exact16: generator AND discriminator on split-operand f16 MFMA
    d.load_state_dict(M.init_discriminator_state(80 + seed))
    d = d.cuda().train()
    ema = R.EMA(g, 0.999)
    ema.register()
    scaler = None
    if precision != "strict":                        # exact16 keeps gradients as f16 pairs: a fixed loss scale like GradScaler's
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=10 ** 9)
    step = RealESRGANStep(g, d, ema, torch.optim.SGD(g.parameters(), 0.0), torch.optim.SGD(d.parameters(), 0.0), scaler=scaler)
    out = step(torch.from_numpy(z["hr_crop"]).cuda(), torch.from_numpy(z["lr"]).cuda())
    torch.cuda.synchronize()
    for k in ("pixel_loss", "adversarial_loss", "d_loss_hr", "d_loss_sr"):
        assert abs(out[k].item() - float(z[k])) < 1e-5, (k, out[k].item(), float(z[k]))
    gn = torch.stack([p.grad.norm() for p in g.parameters()]).cpu()
    dn = torch.stack([p.grad.norm() for p in d.parameters()]).cpu()
    rg, rd = torch.from_numpy(z["g_grad_norms"]), torch.from_numpy(z["d_grad_norms"])
    eg = ((gn - rg).abs() / rg.clamp_min(1e-12)).max().item()
    ed = ((dn - rd).abs() / rd.clamp_min(1e-12)).max().item()
    with open(os.path.join(diag_dir, f"gan_step_golden_{precision}.json"), "w") as f:
        json.dump({"g_grad_norm_worst_rel": eg, "d_grad_norm_worst_rel": ed}, f)
    # exact16 (three-product weight gradients): every gradient norm of both networks within 1e-3 of the reference's own
    tol_n = 1e-3 if precision == "exact16" else 5e-3
    assert eg < tol_n and ed < tol_n, (eg, ed)
    sd = d.state_dict()
    for k in z.files:
        if k.startswith("uv_"):
            assert torch.allclose(sd[k[3:]].cpu(), torch.from_numpy(z[k]), atol=1e-5), k
        if k.startswith("dg_"):
            ref = torch.from_numpy(z[k])
            got = dict(d.named_parameters())[k[3:]].grad.cpu()
            assert ((got - ref).norm() / ref.norm()).item() < 2e-3, k
