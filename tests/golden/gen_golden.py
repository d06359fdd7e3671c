"""Generate the golden vectors under tests/golden/*.npz from the REFERENCE itself.

Run only in the build container (needs /root/reference; see ref_shim.py for the three stand-in
formulas).  Weights are not stored: they are regenerated from a seed by oracle.model_ref.init_*_state
(pure torch CPU generator, deterministic), loaded into the reference modules with load_state_dict, and
the reference's own forward / autograd produce the stored outputs.  The CPU tests then run the oracle
on the same seeds and compare -- that pins the oracle's arithmetic to the reference's.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shim  # noqa: E402
from oracle import model_ref as M  # noqa: E402

PROBE_KEYS = ["conv1.weight", "conv1.bias", "trunk.0.rdb1.conv3.weight", "trunk.11.rdb2.conv1.weight",
              "trunk.22.rdb3.conv5.weight", "trunk.22.rdb3.conv5.bias", "conv2.weight", "upsampling1.0.weight",
              "upsampling2.0.bias", "conv3.0.weight", "conv4.weight", "conv4.bias"]


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: tuple(v.shape) for k, v in out.items()})


def gen_generator(ref):
    for tag, up, seed, shape in (("x4_a", 4, 101, (1, 3, 24, 24)), ("x4_b", 4, 102, (2, 3, 16, 20)),
                                 ("x2", 2, 103, (1, 3, 32, 32)), ("x1", 1, 104, (1, 3, 32, 32))):
        sd = M.init_generator_state(seed, 3, 3, up, bias_noise=0.02)
        sd["conv4.bias"] = sd["conv4.bias"] + 0.5
        g = ref.Generator(3, 3, up)
        g.load_state_dict(sd)
        g.train()
        gen = torch.Generator().manual_seed(seed)
        x = torch.rand(*shape, generator=gen).requires_grad_(True)
        gw = torch.randn(shape[0], 3, shape[2] * up, shape[3] * up, generator=gen)
        y = g(x)
        (y * gw).sum().backward()
        grads = dict(g.named_parameters())
        arrs = {"x": x.detach(), "gw": gw, "y": y.detach(), "gx": x.grad, "seed": seed, "upscale": up,
                "grad_norms": torch.stack([p.grad.norm() for p in g.parameters()])}
        for k in PROBE_KEYS:
            arrs["g_" + k] = grads[k].grad
        save("generator_" + tag, **arrs)


def gen_blocks(ref):
    torch.manual_seed(7)
    rdb = ref.ResidualDenseBlock(64, 32)
    for p in rdb.parameters():          # exercise the bias path too
        if p.dim() == 1:
            p.data.normal_(0, 0.05)
    x = torch.randn(1, 64, 16, 16).requires_grad_(True)
    gw = torch.randn(1, 64, 16, 16)
    y = rdb(x)
    (y * gw).sum().backward()
    arrs = {"x": x.detach(), "gw": gw, "y": y.detach(), "gx": x.grad}
    for k, v in rdb.state_dict().items():
        arrs["w_" + k] = v
    for k, p in rdb.named_parameters():
        arrs["g_" + k] = p.grad
    save("rdb", **arrs)
    torch.manual_seed(8)
    rrdb = ref.ResidualResidualDenseBlock(64, 32)
    x = torch.randn(1, 64, 12, 12)
    with torch.no_grad():
        y = rrdb(x)
    arrs = {"x": x, "y": y}
    for k, v in rrdb.state_dict().items():
        arrs["w_" + k] = v.half()       # stored as f16 to keep the fixture small; the test casts back
    rrdb.load_state_dict({k: v.half().float() for k, v in rrdb.state_dict().items()})
    with torch.no_grad():
        arrs["y"] = rrdb(x)
    save("rrdb", **arrs)


def gen_init(ref):
    """Reference initialisation under torch.manual_seed(0) (config.py:65): per-tensor std and head values."""
    torch.manual_seed(0)
    g = ref.Generator(3, 3, 4)
    sd = g.state_dict()
    keys = list(sd.keys())
    save("generator_init_seed0", keys=np.array(keys), numel=np.array([v.numel() for v in sd.values()]),
         std=torch.stack([v.float().std() if v.numel() > 1 else v.float().abs().sum() for v in sd.values()]),
         head=torch.stack([v.reshape(-1)[:4] if v.numel() >= 4 else torch.cat([v.reshape(-1), torch.zeros(4 - v.numel())])
                           for v in sd.values()]),
         total_sum=torch.stack([v.double().sum() for v in sd.values()]))
    torch.manual_seed(0)
    d = ref.Discriminator()
    sdd = d.state_dict()
    save("discriminator_init_seed0", keys=np.array(list(sdd.keys())),
         shapes=np.array([str(tuple(v.shape)) for v in sdd.values()]),
         total_sum=torch.stack([v.double().sum() for v in sdd.values()]))


def gen_discriminator(ref, seed=201, name="discriminator"):
    """seed 201: the round-1 case; one LeakyReLU pre-activation of its 8 x 8 level lies within fp32 rounding of zero, so two
    correct evaluations may differ by that mask element (~1e-2 in the affected gradients).  seed 231: chosen (a search over
    seeds 200..239 with the oracle in fp32 and float64: worst gradient distance 7.6e-7) so that NO pre-activation is that
    close -- the case the strict / exact16 gradients are held to 1e-3 on."""
    sd = M.init_discriminator_state(seed)
    d = ref.Discriminator()
    d.load_state_dict(sd)
    d.train()
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(2, 3, 64, 64, generator=gen).requires_grad_(True)
    gw = torch.randn(2, 1, 64, 64, generator=gen)
    arrs = {"x": x.detach(), "gw": gw, "seed": seed}
    for call in range(3):                     # three training-mode calls per GAN step (train_realesrgan.py:479,500,508)
        d.zero_grad()
        if x.grad is not None:
            x.grad = None
        y = d(x)
        (y * gw).sum().backward()
        arrs[f"y{call}"] = y.detach()
        arrs[f"gx{call}"] = x.grad.clone()
        arrs[f"u{call}_up_block1"] = d.state_dict()["up_block1.0.weight_u"].clone()
        arrs[f"v{call}_down_block3"] = d.state_dict()["down_block3.0.weight_v"].clone()
        arrs[f"g{call}_conv1.weight"] = d.conv1.weight.grad.clone()
        # slices keep the fixture small
        arrs[f"g{call}_down_block2.weight_orig"] = dict(d.named_parameters())["down_block2.0.weight_orig"].grad[:4].clone()
        arrs[f"g{call}_conv3.weight_orig"] = dict(d.named_parameters())["conv3.0.weight_orig"].grad[:8].clone()
        arrs[f"g{call}_conv4.bias"] = d.conv4.bias.grad.clone()
    d.eval()
    with torch.no_grad():
        arrs["y_eval"] = d(x.detach())
    save(name, **arrs)


def _subsample(t, cap=4096):
    """A gradient tensor as stored in a fixture: whole when small, else every k-th element of the flattened tensor."""
    f = t.reshape(-1)
    return f.clone() if f.numel() <= cap else f[::(f.numel() + cap - 1) // cap].clone()


def gen_discriminator_all_grads(ref, seed, name):
    """Flip-free cases that do not rest on one searched seed: most seeds have no LeakyReLU pre-activation within
    fp32 rounding of zero (oracle fp32 vs float64 over seeds 240..261: sixteen of twenty-two below 1e-5); these fixtures hold
    EVERY one of the 19 gradient tensors (subsampled to <= 4096 elements each, `_subsample`) of the three training calls."""
    sd = M.init_discriminator_state(seed)
    d = ref.Discriminator()
    d.load_state_dict(sd)
    d.train()
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(2, 3, 64, 64, generator=gen).requires_grad_(True)
    gw = torch.randn(2, 1, 64, 64, generator=gen)
    arrs = {"x": x.detach(), "gw": gw, "seed": seed, "names": np.array([k for k, _ in d.named_parameters()])}
    for call in range(3):
        d.zero_grad()
        if x.grad is not None:
            x.grad = None
        y = d(x)
        (y * gw).sum().backward()
        arrs[f"y{call}"] = y.detach()
        arrs[f"gx{call}"] = x.grad.clone()
        for i, (k, p) in enumerate(d.named_parameters()):
            arrs[f"g{call}_{i}"] = _subsample(p.grad)
            arrs[f"n{call}_{i}"] = p.grad.double().norm()
    save(name, **arrs)


def gen_ema(ref):
    torch.manual_seed(9)
    lin = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 4, 3))
    ema = ref.EMA(lin, 0.999)
    ema.register()
    arrs = {}
    for i, (k, p) in enumerate(lin.named_parameters()):
        arrs[f"p0_{i}"] = p.detach().clone()
    for step in range(3):
        with torch.no_grad():
            for p in lin.parameters():
                p.add_(torch.randn_like(p) * 0.01)
        ema.update()
        for i, (k, p) in enumerate(lin.named_parameters()):
            arrs[f"p{step + 1}_{i}"] = p.detach().clone()
            arrs[f"s{step + 1}_{i}"] = ema.shadow[k].clone()
    save("ema", **arrs)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("imgproc", "dataset", "niqe")):
    ref = ref_shim.load("model")
    gen_init(ref)
    gen_blocks(ref)
    gen_generator(ref)
    gen_discriminator(ref)
    gen_discriminator(ref, 231, "discriminator_flipfree")
    gen_discriminator_all_grads(ref, 240, "discriminator_allgrads_240")
    gen_discriminator_all_grads(ref, 243, "discriminator_allgrads_243")
    gen_ema(ref)


# ---------------------------------------------------------------------------------------------------
# degradation ops (reference imgproc.py) -- run with `python tests/golden/gen_golden.py imgproc`
# ---------------------------------------------------------------------------------------------------
def _img(gen, *shape):
    return torch.round(torch.rand(*shape, generator=gen) * 255) / 255


def gen_imgproc():
    import random as pyrandom
    import torch.nn.functional as F
    ip = ref_shim.load("imgproc")
    gen = torch.Generator().manual_seed(301)
    # USM + filter2d
    x = _img(gen, 2, 3, 72, 64)
    usm = ip.USMSharp(50, 0)
    k7 = torch.rand(1, 7, 7, generator=gen); k7 /= k7.sum()
    k21 = torch.rand(2, 21, 21, generator=gen); k21 /= k21.sum(dim=(1, 2), keepdim=True)
    save("imgproc_filter", x=x, usm=usm(x, 0.5, 10), usm_kernel=usm.kernel, k7=k7, f7=ip.filter2d_torch(x, k7),
         k21=k21, f21=ip.filter2d_torch(x, k21))
    # resize call sites (train_realesrnet.py:288 scale_factor=, :326-329 / :349-351 size=)
    x = _img(gen, 2, 3, 48, 40)
    arrs = {"x": x}
    for mode in ("area", "bilinear", "bicubic"):
        for s in (0.3731, 1.3177):
            arrs[f"{mode}_sf_{s}"] = F.interpolate(x, scale_factor=s, mode=mode)
        arrs[f"{mode}_size_30x27"] = F.interpolate(x, size=(30, 27), mode=mode)
        arrs[f"{mode}_size_12x10"] = F.interpolate(x, size=(12, 10), mode=mode)
    save("imgproc_resize", **arrs)
    # noise with the global torch generator seeded (device RNG of the reference == CPU generator here)
    x = _img(gen, 4, 3, 24, 20)
    arrs = {"x": x}
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        arrs[f"gauss_{seed}"] = ip.random_add_gaussian_noise_torch(x, sigma_range=[1, 30], gray_prob=0.4, clip=True, rounds=False)
        torch.manual_seed(seed)
        arrs[f"poisson_{seed}"] = ip.random_add_poisson_noise_torch(x, scale_range=[0.05, 3], gray_prob=0.4, clip=True, rounds=False)
    torch.manual_seed(4)
    arrs["gauss_nogray"] = ip.random_add_gaussian_noise_torch(x, sigma_range=[1, 25], gray_prob=0.0, clip=True, rounds=False)
    torch.manual_seed(4)
    arrs["poisson_allgray"] = ip.random_add_poisson_noise_torch(x, scale_range=[0.05, 2.5], gray_prob=1.0, clip=True, rounds=False)
    save("imgproc_noise", **arrs)
    # DiffJPEG(False): sizes incl. non-multiples of 16, qualities on both sides of 50
    jpeg = ip.DiffJPEG(False)
    arrs = {}
    for tag, (h, w) in (("48x40", (48, 40)), ("77x77", (77, 77)), ("100x100", (100, 100))):
        x = _img(gen, 4, 3, h, w)
        q = torch.tensor([30.0, 49.9, 50.0, 95.0])
        arrs[f"x_{tag}"], arrs[f"q_{tag}"] = x, q.clone()
        arrs[f"y_{tag}"] = jpeg(x, q.clone())                      # forward mutates its quality argument
        factor = torch.tensor([ip._calculate_quality_factor(float(v)) for v in q])
        hp, wp = (16 - h % 16) % 16, (16 - w % 16) % 16
        yq, cbq, crq = jpeg.compress(F.pad(x, (0, wp, 0, hp)), factor)
        arrs[f"cy_{tag}"], arrs[f"ccb_{tag}"], arrs[f"ccr_{tag}"] = yq, cbq, crq
        arrs[f"factor_{tag}"] = factor
    q = torch.tensor([30.0, 49.9, 50.0, 95.0])
    jpeg(arrs["x_48x40"], q)
    arrs["q_mutated"] = q                                           # quirk: caller's tensor now holds the factors
    arrs["y_scalar_q70"] = jpeg(arrs["x_48x40"], 70)
    save("imgproc_jpeg", **arrs)
    # quantise + crop
    lr = torch.rand(2, 3, 25, 25, generator=gen)
    hr = _img(gen, 2, 3, 100, 100)
    pyrandom.seed(5)
    st = pyrandom.getstate()
    top, left = pyrandom.randint(0, 100 - 64), pyrandom.randint(0, 100 - 64)
    pyrandom.setstate(st)
    plr, phr = ip.random_crop(lr, hr, 64, 4)
    save("imgproc_crop", lr=lr, hr=hr, top=top, left=left, plr=plr, phr=phr)
    # host kernel synthesis: fixed-parameter kernels of every type + seeded random draws
    arrs = {}
    for iso in (True, False):
        t = "iso" if iso else "aniso"
        arrs[f"gauss_{t}"] = ip._generate_bivariate_gaussian_kernel(21, 2.1, 0.9, 0.7, isotropic=iso)
        arrs[f"general_{t}"] = ip._generate_bivariate_generalized_gaussian_kernel(15, 1.7, 0.6, -1.1, 2.3, isotropic=iso)
        arrs[f"plateau_{t}"] = ip._generate_bivariate_plateau_gaussian_kernel(9, 2.6, 1.2, 2.0, 1.4, isotropic=iso)
    arrs["sinc_7"] = ip.generate_sinc_kernel(2.5, 7, 0)
    arrs["sinc_13_pad21"] = ip.generate_sinc_kernel(1.1, 13, 21)
    cfg = ref_shim.load("config")
    P = cfg.degradation_model_parameters_dict
    for seed in range(8):
        pyrandom.seed(seed)
        np.random.seed(seed)
        arrs[f"mixed_{seed}"] = ip.random_mixed_kernels(P["gaussian_kernel_type"], P["gaussian_kernel_probability1"], 7 + 2 * seed,
                                                        P["gaussian_sigma_range1"], P["gaussian_sigma_range1"], [-np.pi, np.pi],
                                                        P["generalized_kernel_beta_range1"], P["plateau_kernel_beta_range1"],
                                                        noise_range=None)
    save("imgproc_kernels", **arrs)


def gen_dataset():
    """Host data path (reference dataset.py:64-160, imgproc.py:1599-1687, 1871-2001): full `__getitem__` outputs of the
    reference for a committed PNG under fixed seeds, plus `image_resize` cases."""
    import random
    from PIL import Image
    rimg = ref_shim.load("imgproc")
    rds = ref_shim.load("dataset")
    rcfg = ref_shim.load("config")
    rng = np.random.default_rng(7)
    d = os.path.join(HERE, "dataset_images")
    os.makedirs(d, exist_ok=True)
    png = os.path.join(d, "sample_38x30.png")
    Image.fromarray((rng.random((38, 30, 3)) * 255).astype(np.uint8)).save(png)
    out = {}
    ds = rds.TrainValidImageDataset(d, 16, 4, "Train", rcfg.degradation_model_parameters_dict)
    for seed in range(6):
        random.seed(seed); np.random.seed(seed)
        item = ds[0]
        for k, v in item.items():
            out[f"train{seed}_{k}"] = v.numpy()
    dv = rds.TrainValidImageDataset(d, 16, 4, "Valid", rcfg.degradation_model_parameters_dict)
    item = dv[0]
    out["valid_lr"], out["valid_hr"] = item["lr"].numpy(), item["hr"].numpy()
    for i, (h, w, s) in enumerate([(37, 45, 0.25), (40, 40, 2.0), (33, 50, 0.3), (16, 21, 1.5)]):
        img = rng.random((h, w, 3), dtype=np.float32)
        out[f"resize{i}_in"], out[f"resize{i}_scale"] = img, np.float64(s)
        out[f"resize{i}_out"] = rimg.image_resize(img.copy(), s)
    x = torch.from_numpy(rng.random((2, 3, 8, 9), dtype=np.float32))
    out["ycbcr_in"] = x.numpy()
    out["ycbcr_y"] = rimg.rgb2ycbcr_torch(x.clone(), True).numpy()
    out["ycbcr_full"] = rimg.rgb2ycbcr_torch(x.clone(), False).numpy()
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("wrote dataset.npz", len(out), "arrays")


def gen_niqe():
    """Reference NIQE (image_quality_assessment.py:1001-1032) on stored 8-bit images."""
    import torch.nn.functional as F
    riqa = ref_shim.load("image_quality_assessment")
    path = os.path.join(HERE, "niqe_model.mat")    # copy of the reference's results/pretrained_models/niqe_model.mat
    out = {}
    for i, (seed, b, h, w, cb) in enumerate([(0, 1, 296, 296, 4), (1, 2, 200, 296, 0)]):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(b, 3, h // 4, w // 4, generator=g)
        x = F.interpolate(x, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)
        x = (x * 0.8 + 0.2 * torch.rand(b, 3, h, w, generator=g)).clamp(0, 1)
        u8 = (x * 255).round().to(torch.uint8)
        out[f"img{i}"] = u8.numpy()
        out[f"crop{i}"] = np.int64(cb)
        out[f"score{i}"] = np.atleast_1d(riqa.NIQE(cb, path)(u8.float() / 255.0).numpy())
    np.savez_compressed(os.path.join(HERE, "niqe.npz"), **out)
    print("wrote niqe.npz", {k: v for k, v in out.items() if k.startswith("score")})


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "imgproc":
    gen_imgproc()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "niqe":
    gen_niqe()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "dataset":
    gen_dataset()
