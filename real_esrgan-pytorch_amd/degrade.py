"""The on-the-fly second-order degradation stage (reference train_realesrnet.py:262-377, duplicated at
train_realesrgan.py:342-457) as a HIP pre-processing stage that runs on a side stream and overlaps the
generator step -- the slot the reference's CUDAPrefetcher (dataset.py:271-312) occupies.

All *host* randomness of one batch (which ops fire, resize modes/scales, crop offsets, the 3 x B blur
kernels) is drawn up-front into a `DegradationPlan`, in the reference's draw order from `random` and
`np.random`; all *device* randomness (per-sample sigma / scale / quality, noise fields) is drawn on the
device.  Nothing in the stage synchronises with the host (the reference syncs ~2B+2 times per step).
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib, imgproc
from .config import degradation_model_parameters_dict as MODEL_P
from .config import degradation_process_parameters_dict as PROC_P


@dataclass
class DegradationPlan:
    blur1: bool
    resize1_scale: float
    resize1_mode: str
    noise1_gaussian: bool
    blur2: bool
    resize2_scale: float
    resize2_mode: str
    noise2_gaussian: bool
    sinc_before_jpeg: bool
    resize3_mode: str
    hr_top: int
    hr_left: int
    kernel1: np.ndarray = field(repr=False, default=None)    # [B,21,21] float32
    kernel2: np.ndarray = field(repr=False, default=None)
    sinc_kernel: np.ndarray = field(repr=False, default=None)


def sample_blur_kernels(P: dict = MODEL_P) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """The three kernels of ONE sample, reference dataset.py:82-143 (same draw order)."""
    outs = []
    kr = P["gaussian_kernel_range"]
    for tag in ("1", "2"):
        size = random.choice(kr)
        if np.random.uniform() < P["sinc_kernel_probability" + tag]:
            lo = np.pi / 3 if size < int(np.median(kr)) else np.pi / 5
            k = imgproc.generate_sinc_kernel(np.random.uniform(lo, np.pi), size, padding=False)
        else:
            k = imgproc.random_mixed_kernels(P["gaussian_kernel_type"], P["gaussian_kernel_probability" + tag], size,
                                             P["gaussian_sigma_range" + tag], P["gaussian_sigma_range" + tag],
                                             [-math.pi, math.pi], P["generalized_kernel_beta_range" + tag],
                                             P["plateau_kernel_beta_range" + tag], noise_range=None)
        pad = (kr[-1] - size) // 2
        outs.append(np.pad(k, ((pad, pad), (pad, pad))))
    if np.random.uniform() < P["sinc_kernel_probability3"]:
        size = random.choice(kr)
        s = imgproc.generate_sinc_kernel(np.random.uniform(np.pi / 3, np.pi), size, padding=P["sinc_kernel_size"])
    else:
        s = np.zeros((P["sinc_kernel_size"],) * 2)
        s[P["sinc_kernel_size"] // 2, P["sinc_kernel_size"] // 2] = 1
    return outs[0].astype(np.float32), outs[1].astype(np.float32), s.astype(np.float32)


def _updown(prob, rng):
    kind = random.choices(["up", "down", "keep"], prob)[0]
    if kind == "up":
        return np.random.uniform(1, rng[1])
    if kind == "down":
        return np.random.uniform(rng[0], 1)
    return 1


def sample_plan(batch: int, hr_h: int, hr_w: int, crop: int, P: dict = PROC_P, with_kernels: bool = True) -> DegradationPlan:
    """Host draws of one batch in the reference's order (train_realesrnet.py:275-377)."""
    kernels = [sample_blur_kernels() for _ in range(batch)] if with_kernels else None
    blur1 = np.random.uniform() <= P["first_blur_probability"]                    # :275
    s1 = _updown(P["resize_probability1"], P["resize_range1"])                    # :279-286
    m1 = random.choice(["area", "bilinear", "bicubic"])                           # :287
    g1 = np.random.uniform() < P["gaussian_noise_probability1"]                   # :291
    blur2 = np.random.uniform() < P["second_blur_probability"]                    # :313
    s2 = _updown(P["resize_probability2"], P["resize_range2"])                    # :317-324
    m2 = random.choice(["area", "bilinear", "bicubic"])                           # :325
    g2 = np.random.uniform() < P["gaussian_noise_probability2"]                   # :332
    first = np.random.uniform() < 0.5                                             # :347
    m3 = random.choice(["area", "bilinear", "bicubic"])                           # :351 / :368
    top = random.randint(0, hr_h - crop)                                          # imgproc.py:1913
    left = random.randint(0, hr_w - crop)                                         # imgproc.py:1914
    plan = DegradationPlan(bool(blur1), float(s1), m1, bool(g1), bool(blur2), float(s2), m2, bool(g2), bool(first), m3, top, left)
    if kernels is not None:
        plan.kernel1 = np.stack([k[0] for k in kernels])
        plan.kernel2 = np.stack([k[1] for k in kernels])
        plan.sinc_kernel = np.stack([k[2] for k in kernels])
    return plan


def _noise_stage(out, gaussian: bool, sigma_range, scale_range, gray_prob, inj: Optional[dict]):
    """train_realesrnet.py:291-304 / 332-345.  `inj` (parity tests): the reference's own device draws for this stage --
    Gaussian: sigma[B], gray[B], field_gray[h,w] or None, field_color[B,3,h,w] (imgproc.py:933-936, 854, 858); Poisson:
    scale[B], gray[B] (the Poisson samples themselves come from the device's Philox stream and cannot be injected)."""
    if inj is None:
        if gaussian:
            return imgproc.random_add_gaussian_noise_torch(out, sigma_range, gray_prob, True, False)
        return imgproc.random_add_poisson_noise_torch(out, scale_range, gray_prob, True, False)
    if gaussian:
        return imgproc.add_gaussian_noise_fields(out, inj["sigma"], inj["gray"], inj.get("field_gray"), inj["field_color"], True, False)
    return imgproc.add_poisson_noise(out, inj["scale"], inj["gray"], inj.get("seed", 1), True, False)


def run_plan(hr: torch.Tensor, plan: DegradationPlan, usm: imgproc.USMSharp, jpeg: imgproc.DiffJPEG, upscale: int,
             crop: int, P: dict = PROC_P, trace: Optional[dict] = None, inject: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Execute one plan on the current stream.  `trace` (tests) receives every intermediate.  `inject` (parity tests)
    replaces device draws -- keys "noise1"/"noise2" (see _noise_stage), "q1"/"q2" (JPEG qualities [B]) -- and can
    substitute a stage's output before the next stage runs: "replace": {stage name: tensor} (used after a Poisson
    stage, whose samples cannot be matched draw for draw)."""
    dev = hr.device
    b, _, H, W = hr.shape
    inject = inject or {}
    replace = inject.get("replace", {})

    def rec(name, t):
        if trace is not None:
            trace[name] = t
        return replace.get(name, t)

    def dev_kernel(k):   # device tensors (the prefetchers upload them ahead), numpy from the plan's own sampler, or host tensors of a dataset batch
        k = k if torch.is_tensor(k) else torch.from_numpy(k)
        if not k.is_cuda and not k.is_pinned():
            k = k.pin_memory()             # a copy from pageable memory would block the host until the stream reaches it
        return k.to(dev, non_blocking=True)

    k1, k2, ks = dev_kernel(plan.kernel1), dev_kernel(plan.kernel2), dev_kernel(plan.sinc_kernel)
    out = rec("usm", usm(hr, 0.5, 10))                                                              # :268
    if plan.blur1:
        out = rec("blur1", imgproc.filter2d_torch(out, k1))                                         # :276
    out = rec("resize1", imgproc.interpolate(out, scale_factor=plan.resize1_scale, mode=plan.resize1_mode))   # :288
    out = rec("noise1", _noise_stage(out, plan.noise1_gaussian, P["noise_range1"], P["poisson_scale_range1"],  # :291-304
                                     P["gray_noise_probability1"], inject.get("noise1")))
    q = inject["q1"].to(dev) if "q1" in inject else torch.empty(b, device=dev).uniform_(*P["jpeg_range1"])   # :307
    out = rec("jpeg1", jpeg(out, q, clamp_input=True))                                                    # :308-309
    if plan.blur2:
        out = rec("blur2", imgproc.filter2d_torch(out, k2))                                         # :314
    size2 = (int(H / upscale * plan.resize2_scale), int(W / upscale * plan.resize2_scale))         # :327-328
    out = rec("resize2", imgproc.interpolate(out, size=size2, mode=plan.resize2_mode))
    out = rec("noise2", _noise_stage(out, plan.noise2_gaussian, P["noise_range2"], P["poisson_scale_range2"],  # :332-345
                                     P["gray_noise_probability2"], inject.get("noise2")))
    size3 = (H // upscale, W // upscale)
    q2 = inject["q2"].to(dev) if "q2" in inject else torch.empty(b, device=dev).uniform_(*P["jpeg_range2"])
    if plan.sinc_before_jpeg:                                                                       # :347-358
        out = rec("resize3", imgproc.interpolate(out, size=size3, mode=plan.resize3_mode))
        out = rec("sinc", imgproc.filter2d_torch(out, ks))
        out = rec("jpeg2", jpeg(out, q2, clamp_input=True))
    else:                                                                                           # :359-371
        out = rec("jpeg2", jpeg(out, q2, clamp_input=True))
        out = rec("resize3", imgproc.interpolate(out, size=size3, mode=plan.resize3_mode))
        out = rec("sinc", imgproc.filter2d_torch(out, ks))
    rec("final", out)
    return imgproc.quantize_crop(out, hr, crop, upscale, plan.hr_top, plan.hr_left)                 # :374-377


class Degrader:
    """Prefetching degradation stage.  `__call__(hr)` returns the (lr, hr_crop) pair computed for the
    previous submission and immediately enqueues the degradation of `hr` on the side stream, so it runs
    under the generator's forward/backward (one batch of latency, exactly like CUDAPrefetcher.next()).
    The per-sample blur / sinc kernels are drawn inline on the calling thread (~0.2 ms per image on the GPU box's host: the
    enqueue runs ahead of the device; feeding them from DataLoader workers was measured 5 % SLOWER at 32 images per 17 ms step)."""

    def __init__(self, batch: int, hr_size: int, upscale: int = 4, crop: int = 256, seed: int = 0,
                 device: Optional[torch.device] = None) -> None:
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.batch, self.hr_size, self.upscale, self.crop = batch, hr_size, upscale, crop
        self.stream = torch.cuda.Stream(device=self.device)
        self.usm = imgproc.USMSharp(50, 0).to(self.device)           # train_realesrnet.py:234
        self.jpeg = imgproc.DiffJPEG(False)                           # train_realesrnet.py:231
        random.seed(seed)
        np.random.seed(seed)
        self._pending = None

    def _enqueue(self, hr: torch.Tensor):
        plan = sample_plan(self.batch, hr.shape[2], hr.shape[3], self.crop)
        # The blur kernels go up FIRST, from pinned memory, before the side stream is told to wait for the main stream: a copy from
        # pageable memory blocks the host until the stream reaches it -- i.e. until everything the main stream had queued is done --
        # which drained the device once per step (measured: 6 ms of idle device in front of every degradation at 32 x 64^2).
        with torch.cuda.stream(self.stream):
            for name in ("kernel1", "kernel2", "sinc_kernel"):
                setattr(plan, name, torch.from_numpy(getattr(plan, name)).pin_memory().to(self.device, non_blocking=True))
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            lr, hrc = run_plan(hr, plan, self.usm, self.jpeg, self.upscale, self.crop)
            # An identity HR window comes back as `hr` itself (imgproc.quantize_crop).  The pair is handed out one call LATER: a caller that
            # refills its HR buffer in place in between would get LR(i) next to HR(i + 1), so the pair keeps its own copy (0.4 GB of the
            # stage's 3.4 GB per batch at the headline geometry); direct run_plan callers keep the alias.
            if hrc.data_ptr() == hr.data_ptr():
                hrc = hr.clone()
        done = torch.cuda.Event()
        done.record(self.stream)
        return lr, hrc, done

    def __call__(self, hr: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        _lib.require_cuda(hr, "Degrader")
        if self._pending is None:
            self._pending = self._enqueue(hr)
        lr, hrc, done = self._pending
        torch.cuda.current_stream().wait_event(done)
        lr.record_stream(torch.cuda.current_stream())
        hrc.record_stream(torch.cuda.current_stream())
        self._pending = self._enqueue(hr)          # next batch degrades under this batch's generator step
        return lr, hrc


class DegradationPrefetcher:
    """What the two training scripts iterate over: the reference's batch source (`CUDAPrefetcher`, dataset.py:271-312) with
    the degradation stage of train_realesrnet.py:262-377 attached ONE BATCH AHEAD on a side stream -- the device-side
    counterpart of the prefetcher's upload slot.  `next()` returns (lr, hr_crop, batch) of the batch whose degradation was
    enqueued during the previous call and, before returning, takes the following batch from the source and enqueues ITS
    degradation on the side stream: those kernels run under the generator step the caller is about to issue.  The blur
    kernels are the ones the dataset sampled per image (dataset.py:82-143); the host draws of a batch (`sample_plan`) happen
    in the reference's order -- source batch i+1, then plan i -- so a fixed seed gives the reference's sequence.

    `done_events` keeps the (start, end) events of the last enqueued degradation for tests."""

    def __init__(self, source, usm: imgproc.USMSharp, jpeg: imgproc.DiffJPEG, upscale: int, crop: int,
                 device: Optional[torch.device] = None) -> None:
        self.source, self.usm, self.jpeg, self.upscale, self.crop = source, usm, jpeg, upscale, crop
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self._pending = None
        self.done_events = None

    def __len__(self) -> int:
        return len(self.source)

    def _enqueue(self, batch):
        if batch is None:
            return None
        hr = batch["hr"].to(device=self.device, non_blocking=True)
        _lib.require_cuda(hr, "DegradationPrefetcher")
        plan = sample_plan(hr.shape[0], hr.shape[2], hr.shape[3], self.crop, with_kernels=False)
        plan.kernel1, plan.kernel2, plan.sinc_kernel = batch["kernel1"], batch["kernel2"], batch["sinc_kernel"]
        main = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(main)              # the upload of `batch` was ordered into the consumer stream by the source
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            start.record(self.stream)
            lr, hrc = run_plan(hr, plan, self.usm, self.jpeg, self.upscale, self.crop)
            # An identity HR window comes back as `hr` itself (imgproc.quantize_crop).  The pair is handed out one call LATER: a caller that
            # refills its HR buffer in place in between would get LR(i) next to HR(i + 1), so the pair keeps its own copy (0.4 GB of the
            # stage's 3.4 GB per batch at the headline geometry); direct run_plan callers keep the alias.
            if hrc.data_ptr() == hr.data_ptr():
                hrc = hr.clone()
            end.record(self.stream)
        for t in (hr, batch["kernel1"], batch["kernel2"], batch["sinc_kernel"]):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.stream)
        self.done_events = (start, end)
        return lr, hrc, batch, end

    def reset(self) -> None:
        self.source.reset()
        self._pending = self._enqueue(self.source.next())

    def next(self):
        if self._pending is None:
            return None
        lr, hrc, batch, done = self._pending
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        lr.record_stream(main)
        hrc.record_stream(main)
        self._pending = self._enqueue(self.source.next())     # batch i+1 degrades under step i
        return lr, hrc, batch
