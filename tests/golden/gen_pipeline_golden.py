"""Golden vectors of the WHOLE second-order degradation + one RealESRNet step, produced by executing the
reference's own `train()` (train_realesrnet.py:209-413) for one batch on the CPU.

Run only in the build container (needs /root/reference; stand-ins of ref_shim.py).  What is recorded:
  * the batch handed to the loop (HR tiles, the three blur kernels per sample);
  * every HOST decision of the loop body in draw order (np.random / random; train_realesrnet.py:275-351 and
    imgproc.py:1913-1914), by seeding `random`, `np.random` and torch exactly like config.py:64-66;
  * every DEVICE draw (torch.rand / torch.randn / torch.poisson / Tensor.uniform_ as called by imgproc.py:829-964
    and train_realesrnet.py:307,355,361), in call order, so the oracle and the HIP path can be fed the same fields;
  * every intermediate image (after USM, blur, resize, noise, JPEG, ... , quantise + crop);
  * the L1 loss of the step and the gradient norm of each of the 702 parameter tensors.
Weights are regenerated from a seed by oracle.model_ref.init_generator_state and loaded into the reference
Generator with load_state_dict.

    python tests/golden/gen_pipeline_golden.py [pipeline|gan]   # rewrites tests/golden/pipeline_*.npz / gan_step_*.npz
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shim  # noqa: E402
from oracle import imgproc_ref as I  # noqa: E402
from oracle import model_ref as M  # noqa: E402

HR_SIZE, CROP, BATCH = 88, 64, 2


def _load_train_module():
    ref_shim.install_stubs()
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = object
        sys.modules["torch.utils.tensorboard"] = tb
    for name in ("natsort",):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.natsorted = sorted
            sys.modules[name] = m
    return ref_shim.load("train_realesrnet")


class _OneBatch:
    """Stands in for CUDAPrefetcher (dataset.py:271-312): one batch, then None."""

    def __init__(self, batch):
        self.batch, self.done = batch, False

    def reset(self):
        self.done = False

    def next(self):
        if self.done:
            return None
        self.done = True
        return self.batch

    def __len__(self):
        return 1


class _Writer:
    def add_scalar(self, *a, **k):
        pass


def run_case(seed):
    T = _load_train_module()
    cfg, ip = T.config, T.imgproc
    cfg.device = torch.device("cpu")
    cfg.image_size = CROP
    cfg.upscale_factor = 4
    cfg.print_frequency = 1000

    # the batch: quantised HR tiles + per-sample kernels from the (dataset-golden pinned) sampler
    gen = torch.Generator().manual_seed(1000 + seed)
    base = torch.rand(BATCH, 3, HR_SIZE // 8, HR_SIZE // 8, generator=gen)
    hr = torch.nn.functional.interpolate(base, size=(HR_SIZE, HR_SIZE), mode="bicubic").clamp(0, 1)
    hr = torch.round((0.85 * hr + 0.15 * torch.rand(BATCH, 3, HR_SIZE, HR_SIZE, generator=gen)) * 255) / 255
    random.seed(500 + seed)
    np.random.seed(500 + seed)
    ks = [I.sample_sample_kernels(cfg.degradation_model_parameters_dict) for _ in range(BATCH)]
    k1 = torch.from_numpy(np.stack([k[0] for k in ks])).float()
    k2 = torch.from_numpy(np.stack([k[1] for k in ks])).float()
    ksinc = torch.from_numpy(np.stack([k[2] for k in ks])).float()
    batch = {"hr": hr.clone(), "kernel1": k1.clone(), "kernel2": k2.clone(), "sinc_kernel": ksinc.clone()}

    sd = M.init_generator_state(40 + seed, 3, 3, 4, bias_noise=0.02)
    sd["conv4.bias"] = sd["conv4.bias"] + 0.5
    model = T.Generator(3, 3, 4)
    model.load_state_dict(sd)
    ema = T.EMA(model, 0.999)
    ema.register()
    opt = torch.optim.Adam(model.parameters(), 2e-4, (0.9, 0.99))
    crit = torch.nn.L1Loss()

    rec, draws, host = {}, [], []
    counts = {}

    def put(name, t):
        n = counts.get(name, 0)
        counts[name] = n + 1
        rec[f"{name}{n + 1}"] = t.detach().clone()

    # ---- recorders around the reference's ops (the loop's own code runs unchanged) ----
    orig = dict(filter2d=ip.filter2d_torch, interp=torch.nn.functional.interpolate, gauss=ip.random_add_gaussian_noise_torch,
                poisson=ip.random_add_poisson_noise_torch, jpeg=ip.DiffJPEG.forward, usm=ip.USMSharp.forward,
                crop=ip.random_crop, rand=torch.rand, randn=torch.randn, tpoisson=torch.poisson,
                uniform_=torch.Tensor.uniform_, np_uniform=np.random.uniform, choices=random.choices, choice=random.choice,
                randint=random.randint)
    in_model = [False]

    def w_filter(image, kernel):
        out = orig["filter2d"](image, kernel)
        put("filter", out)
        return out

    def w_interp(x, *a, **k):
        out = orig["interp"](x, *a, **k)
        if not in_model[0]:
            put("resize", out)
        return out

    def w_gauss(*a, **k):
        host.append(("noise", "gaussian"))
        out = orig["gauss"](*a, **k)
        put("noise", out)
        return out

    def w_poisson(*a, **k):
        host.append(("noise", "poisson"))
        out = orig["poisson"](*a, **k)
        put("noise", out)
        return out

    def w_jpeg(self, x, quality):
        put("jpeg_q", quality)
        # .contiguous(): under torch 2.10 the reference's DiffJPEG returns a permuted view and its own
        # filter2d_torch (imgproc.py:1116, `.view`) then raises; values are unchanged by the copy
        out = orig["jpeg"](self, x, quality).contiguous()
        put("jpeg", out)
        return out

    def w_usm(self, x, weight, threshold):
        out = orig["usm"](self, x, weight, threshold)
        put("usm", out)
        return out

    def w_crop(lr, hr_, size, up):
        put("lr_full", lr)
        a, b = orig["crop"](lr, hr_, size, up)
        put("lr", a)
        put("hr_crop", b)
        return a, b

    def w_rand(*a, **k):
        out = orig["rand"](*a, **k)
        if not in_model[0]:
            draws.append(("rand", out.clone()))
        return out

    def w_randn(*a, **k):
        out = orig["randn"](*a, **k)
        if not in_model[0]:
            draws.append(("randn", out.clone()))
        return out

    def w_tpoisson(x, *a, **k):
        out = orig["tpoisson"](x, *a, **k)
        draws.append(("poisson", out.clone()))
        return out

    def w_uniform_(self, *a, **k):
        out = orig["uniform_"](self, *a, **k)
        if not in_model[0]:
            draws.append(("uniform", out.clone()))
        return out

    def w_np_uniform(*a, **k):
        v = orig["np_uniform"](*a, **k)
        host.append(("np.uniform", float(v)))
        return v

    def w_choices(*a, **k):
        v = orig["choices"](*a, **k)
        host.append(("choices", v[0]))
        return v

    def w_choice(seq):
        v = orig["choice"](seq)
        host.append(("choice", v))
        return v

    def w_randint(a, b):
        v = orig["randint"](a, b)
        host.append(("randint", int(v)))
        return v

    real_forward = model.forward

    def model_forward(x):
        in_model[0] = True
        try:
            y = real_forward(x)
        finally:
            in_model[0] = False
        rec["sr"] = y.detach().clone()
        return y

    model.forward = model_forward
    real_crit = crit.forward

    def crit_forward(a, b):
        v = real_crit(a, b)
        rec["loss"] = v.detach().clone()
        return v

    crit.forward = crit_forward

    # config.py:64-66: the three host generators share one seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    ip.filter2d_torch, torch.nn.functional.interpolate = w_filter, w_interp
    ip.random_add_gaussian_noise_torch, ip.random_add_poisson_noise_torch = w_gauss, w_poisson
    ip.DiffJPEG.forward, ip.USMSharp.forward, ip.random_crop = w_jpeg, w_usm, w_crop
    torch.rand, torch.randn, torch.poisson, torch.Tensor.uniform_ = w_rand, w_randn, w_tpoisson, w_uniform_
    np.random.uniform, random.choices, random.choice, random.randint = w_np_uniform, w_choices, w_choice, w_randint
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scaler = T.amp.GradScaler()
            T.train(model, ema, _OneBatch(batch), crit, opt, 0, scaler, _Writer())
    finally:
        ip.filter2d_torch, torch.nn.functional.interpolate = orig["filter2d"], orig["interp"]
        ip.random_add_gaussian_noise_torch, ip.random_add_poisson_noise_torch = orig["gauss"], orig["poisson"]
        ip.DiffJPEG.forward, ip.USMSharp.forward, ip.random_crop = orig["jpeg"], orig["usm"], orig["crop"]
        torch.rand, torch.randn, torch.poisson, torch.Tensor.uniform_ = orig["rand"], orig["randn"], orig["tpoisson"], orig["uniform_"]
        np.random.uniform, random.choices, random.choice, random.randint = orig["np_uniform"], orig["choices"], orig["choice"], orig["randint"]

    grad_norms = torch.stack([p.grad.norm() for p in model.parameters()])
    return dict(hr=hr, k1=k1, k2=k2, ksinc=ksinc, rec=rec, draws=draws, host=host, grad_norms=grad_norms,
                probe={k: dict(model.named_parameters())[k].grad.clone() for k in
                       ("conv1.weight", "trunk.11.rdb2.conv3.weight", "conv4.bias")})


def run_gan_case(seed):
    """One batch through the reference's RealESRGAN `train()` (train_realesrgan.py:282-553) on the CPU: the LR/HR pair its
    degradation produced, SR, the four loss values, the gradient norms of every generator and discriminator tensor, and the
    spectral-norm vectors after the step's three discriminator calls.  The VGG19 criterion is a stand-in returning zeros
    (torchvision and its weights are absent; in the reference the term is detached anyway, :477-478)."""
    _load_train_module()
    T = ref_shim.load("train_realesrgan")
    cfg, ip = T.config, T.imgproc
    cfg.device = torch.device("cpu")
    cfg.image_size, cfg.upscale_factor, cfg.print_frequency = CROP, 4, 1000
    cfg.pixel_weight, cfg.content_weight, cfg.adversarial_weight = 1.0, [0.1, 0.1, 1.0, 1.0, 1.0], 0.1     # config.py:135-138
    gen = torch.Generator().manual_seed(2000 + seed)
    base = torch.rand(BATCH, 3, HR_SIZE // 8, HR_SIZE // 8, generator=gen)
    hr = torch.nn.functional.interpolate(base, size=(HR_SIZE, HR_SIZE), mode="bicubic").clamp(0, 1)
    hr = torch.round((0.85 * hr + 0.15 * torch.rand(BATCH, 3, HR_SIZE, HR_SIZE, generator=gen)) * 255) / 255
    random.seed(700 + seed)
    np.random.seed(700 + seed)
    ks = [I.sample_sample_kernels(cfg.degradation_model_parameters_dict) for _ in range(BATCH)]
    batch = {"hr": hr.clone(), "kernel1": torch.from_numpy(np.stack([k[0] for k in ks])).float(),
             "kernel2": torch.from_numpy(np.stack([k[1] for k in ks])).float(),
             "sinc_kernel": torch.from_numpy(np.stack([k[2] for k in ks])).float()}
    gsd = M.init_generator_state(60 + seed, 3, 3, 4, bias_noise=0.02)
    gsd["conv4.bias"] = gsd["conv4.bias"] + 0.5
    dsd = M.init_discriminator_state(80 + seed)
    g = T.Generator(3, 3, 4)
    g.load_state_dict(gsd)
    d = T.Discriminator()
    d.load_state_dict(dsd)
    ema = T.EMA(g, 0.999)
    ema.register()
    g_opt = torch.optim.Adam(g.parameters(), 1e-4, (0.9, 0.99))
    d_opt = torch.optim.Adam(d.parameters(), 1e-4, (0.9, 0.99))
    pixel, adv = torch.nn.L1Loss(), torch.nn.BCEWithLogitsLoss()
    rec = {"adv": []}

    class ZeroContent(torch.nn.Module):
        def forward(self, sr, hr_):
            return (0.0, 0.0, 0.0, 0.0, 0.0)

    orig_crop, orig_jpeg = ip.random_crop, ip.DiffJPEG.forward

    def w_crop(lr_, hr_, size, up):
        a, b = orig_crop(lr_, hr_, size, up)
        rec["lr"], rec["hr_crop"] = a.detach().clone(), b.detach().clone()
        return a, b

    def w_jpeg(self, x, quality):
        return orig_jpeg(self, x, quality).contiguous()      # see run_case

    real_g = g.forward

    def g_forward(x):
        y = real_g(x)
        rec["sr"] = y.detach().clone()
        return y
    g.forward = g_forward
    real_pixel, real_adv = pixel.forward, adv.forward

    def pixel_forward(a, b):
        v = real_pixel(a, b)
        rec["pixel"] = v.detach().clone()
        return v

    def adv_forward(a, b):
        v = real_adv(a, b)
        rec["adv"].append(v.detach().clone())
        return v
    pixel.forward, adv.forward = pixel_forward, adv_forward
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    ip.random_crop, ip.DiffJPEG.forward = w_crop, w_jpeg
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            T.train(d, g, ema, _OneBatch(batch), pixel, ZeroContent(), adv, d_opt, g_opt, 0, T.amp.GradScaler(), _Writer())
    finally:
        ip.random_crop, ip.DiffJPEG.forward = orig_crop, orig_jpeg
    assert len(rec["adv"]) == 3
    out = {"seed": seed, "lr": rec["lr"], "hr_crop": rec["hr_crop"], "sr": rec["sr"], "pixel_loss": rec["pixel"],
           "adversarial_loss": 0.1 * rec["adv"][0], "d_loss_hr": rec["adv"][1], "d_loss_sr": rec["adv"][2],
           "g_grad_norms": torch.stack([p.grad.norm() for p in g.parameters()]),
           "d_grad_norms": torch.stack([p.grad.norm() for p in d.parameters()])}
    for k, v in d.state_dict().items():
        if k.endswith("weight_u") or k.endswith("weight_v"):
            out["uv_" + k] = v.clone()
    dn = dict(d.named_parameters())
    for k in ("conv1.weight", "conv3.0.weight_orig", "conv4.bias"):
        out["dg_" + k] = dn[k].grad.clone()
    return out


def describe(host):
    """The loop body's host decisions in plan form (same fields as degrade.DegradationPlan)."""
    it = iter(host)
    d = {}

    def nxt(kind):
        k, v = next(it)
        assert k == kind, (k, kind)
        return v
    d["blur1"] = nxt("np.uniform") <= 1.0                                      # first_blur_probability = 1.0 (config.py:44)
    ud = nxt("choices")
    d["resize1_scale"] = nxt("np.uniform") if ud != "keep" else 1.0
    d["resize1_mode"] = nxt("choice")
    u = nxt("np.uniform")
    d["noise1"] = nxt("noise")
    d["noise1_u"] = u
    d["blur2_u"] = nxt("np.uniform")
    ud = nxt("choices")
    d["resize2_scale"] = nxt("np.uniform") if ud != "keep" else 1.0
    d["resize2_mode"] = nxt("choice")
    u = nxt("np.uniform")
    d["noise2"] = nxt("noise")
    d["noise2_u"] = u
    d["sinc_first_u"] = nxt("np.uniform")
    d["resize3_mode"] = nxt("choice")
    d["hr_top"] = nxt("randint")
    d["hr_left"] = nxt("randint")
    rest = list(it)
    assert not rest, rest
    return d


def main():
    want = {("gaussian", "poisson"), ("poisson", "gaussian")}     # both noise kinds in both positions
    got = {}
    seed = 0
    sinc_seen = set()                                             # ... and both orders of the final resize/sinc/JPEG block
    while want and seed < 64:
        case = run_case(seed)
        d = describe(case["host"])
        key = (d["noise1"], d["noise2"])
        if key in want and d["blur2_u"] < 0.8 and (d["sinc_first_u"] < 0.5) not in sinc_seen:
            want.discard(key)
            sinc_seen.add(d["sinc_first_u"] < 0.5)
            got[seed] = (case, d)
        seed += 1
    assert not want, want
    P = _load_train_module().config.degradation_process_parameters_dict
    rename = {"filter3": "blur1", "filter4": "blur2", "filter5": "sinc", "usm1": "usm", "lr_full1": "lr_full", "lr1": "lr",
              "hr_crop1": "hr_crop", "jpeg_q1": "q1", "jpeg_q2": "q2"}
    for seed, (case, d) in got.items():
        arrs = {"seed": seed, "hr": case["hr"], "k1": case["k1"], "k2": case["k2"], "ksinc": case["ksinc"],
                "grad_norms": case["grad_norms"]}
        # the decisions in plan form (oracle.degrade_ref.sample_plan / the product's degrade.sample_plan field names)
        plan = {"blur1": d["blur1"], "resize1_scale": d["resize1_scale"], "resize1_mode": d["resize1_mode"],
                "noise1_gaussian": d["noise1"] == "gaussian", "blur2": d["blur2_u"] < P["second_blur_probability"],
                "resize2_scale": d["resize2_scale"], "resize2_mode": d["resize2_mode"],
                "noise2_gaussian": d["noise2"] == "gaussian", "sinc_before_jpeg": d["sinc_first_u"] < 0.5,
                "resize3_mode": d["resize3_mode"], "hr_top": d["hr_top"], "hr_left": d["hr_left"]}
        assert plan["blur1"] and plan["blur2"]
        assert (d["noise1_u"] < P["gaussian_noise_probability1"]) == plan["noise1_gaussian"]
        assert (d["noise2_u"] < P["gaussian_noise_probability2"]) == plan["noise2_gaussian"]
        for k, v in plan.items():
            arrs["plan_" + k] = np.asarray(v)
        for k, v in case["rec"].items():
            if k in ("filter1", "filter2"):      # the two blurs inside USMSharp.forward (imgproc.py:1527,1531)
                continue
            arrs["t_" + rename.get(k, k)] = v
        for i, (kind, t) in enumerate(case["draws"]):
            arrs[f"draw_{i:02d}_{kind}"] = t
        for k, v in case["probe"].items():
            arrs["g_" + k] = v
        out = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
        path = os.path.join(HERE, f"pipeline_seed{seed}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), plan)
        print("   draws:", [(k, tuple(t.shape)) for k, t in case["draws"]])
        print("   stored:", sorted(k for k in out if k.startswith("t_")))


def main_gan():
    case = run_gan_case(5)
    out = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in case.items()}
    path = os.path.join(HERE, "gan_step_seed5.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), {k: float(out[k]) for k in ("pixel_loss", "adversarial_loss", "d_loss_hr", "d_loss_sr")})


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "pipeline":
        main()
    if len(sys.argv) < 2 or sys.argv[1] == "gan":
        main_gan()
