"""Train steps of the hot path (reference train_realesrnet.py:258-413 and train_realesrgan.py:338-556 loop bodies),
data-parallel ready.

One process per GPU (`setup_distributed`).  Exchanges per optimiser step (`DataParallel`):
  * generator: its backward writes every weight gradient into one flat fp32 arena and fires an event per finished range
    (tail convs, RRDB 22 ... 0, conv1); the ranges are all-reduced bucket by bucket on a communication stream WHILE the
    rest of the backward pass runs (`attach` / `all_reduce_ranges_`);
  * discriminator (GAN step): its gradients are the sum of two backward passes, reduced once after the second
    (`attach_discriminator` / `all_reduce_grads_`); weights and spectral-norm u / v are broadcast at attach time.
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import losses
from .model import EMA, Generator


def setup_distributed(backend: Optional[str] = None) -> tuple:
    """One process per GPU: bind this process to `cuda:LOCAL_RANK` and, when launched with WORLD_SIZE > 1
    (`python -m torch.distributed.run --nproc-per-node N ...`), join the process group -- backend "nccl" is RCCL over xGMI.
    Returns (rank, world, device).  Safe to call in a single-process run (world 1, no process group)."""
    # the HSA runtime reads its environment when the first HIP call initialises it: set it before anything touches the GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or os.environ.get("RESR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def _arena_of(tensors: List[torch.Tensor]) -> Optional[torch.Tensor]:
    """The flat fp32 tensor `tensors` are consecutive views of, if they are (same storage, back to back)."""
    if not tensors or any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        return None
    base = tensors[0]
    off = base.storage_offset()
    for t in tensors:
        if t.untyped_storage().data_ptr() != base.untyped_storage().data_ptr() or t.storage_offset() != off:
            return None
        off += t.numel()
    total = off - base.storage_offset()
    return torch.as_strided(base, (total,), (1,), base.storage_offset())


class DataParallel:
    """Pure data parallelism over the GPUs of one node: replicas hold identical
    weights, every step all-reduces (mean) the flat gradient arena.  With backend "nccl" this is
    RCCL over xGMI; "gloo" serves the CPU tests of the host logic."""

    def __init__(self, bucket_bytes: int = 12 << 20, force: Optional[bool] = None) -> None:
        """force (default $RESR_DP_FORCE=1): run every collective branch even with a world of ONE rank -- a world-1 `nccl` group
        launches real RCCL kernels on one GPU, which is how the single-GPU tests and `RESR_BENCH_FORCE_NCCL=1 bench.py` execute
        the RCCL path (ReduceOp.AVG on arena slices, the communication stream, RCCL kernels next to chained conv launches)."""
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if force is None:
            force = os.environ.get("RESR_DP_FORCE", "0") == "1"
        self.active = self.world > 1 or bool(force and dist.is_initialized())
        # exchange issued once behind the backward pass (overlap off, the default): ONE message -- RCCL chunks and pipelines a
        # large all-reduce itself, and every extra collective is a launch plus a stream join (a world-1 RCCL group on one GPU:
        # 0.66 ms per RealESRGAN step for eight bucketed collectives).  $RESR_DP_BUCKETED=1: bucket_bytes pieces.
        self.bucketed = os.environ.get("RESR_DP_BUCKETED", "0") == "1"
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.backend = dist.get_backend() if dist.is_initialized() else None
        # RCCL averages inside the collective (ncclAvg); gloo (the CPU tests) sums and the mean is one more pass
        self._avg = self.backend == "nccl"
        self._warm = False

    def _op(self):
        return dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM

    def _finish_mean_(self, flat: torch.Tensor) -> None:
        if not self._avg:
            flat.mul_(1.0 / self.world)

    def warm_up(self, device) -> None:
        """Communicator set-up (seconds on a cold node) happens on the first collective: do it HERE, with every rank
        waiting, not under the first training step -- a chained conv launch next to a collective that waits seconds for a
        late peer would run into its poll time-out (conv3x3_ws.h)."""
        if self.active and not self._warm:
            t = torch.zeros(1024, dtype=torch.float32, device=device)
            dist.all_reduce(t)
            if t.is_cuda:
                torch.cuda.synchronize(device)
            dist.barrier()
            self._warm = True

    def broadcast_(self, flat: torch.Tensor) -> None:
        if self.active:
            dist.broadcast(flat, src=0)

    def all_reduce_mean_(self, flat: torch.Tensor) -> None:
        """Mean over ranks of a whole arena, issued behind the pass that produced it: one message (`bucketed`: pieces of
        bucket_bytes, the form the overlapped exchange `all_reduce_ranges_` uses because its ranges become final one by one)."""
        if not self.active:
            return
        works = []
        step = self.bucket_elems if self.bucketed else max(1, flat.numel())
        for off in range(0, flat.numel(), step):
            works.append(dist.all_reduce(flat[off:off + step], op=self._op(), async_op=True))
        for w in works:
            w.wait()
        self._finish_mean_(flat)

    def attach(self, model: Generator, overlap: Optional[bool] = None) -> None:
        """Generator: identical initial weights, then the gradient arena is all-reduced once per backward (the generator runs
        exactly one backward per optimiser step in both training scripts: train_realesrnet.py:388, train_realesrgan.py:484).
        overlap=True: bucket by bucket on a communication stream WHILE the rest of the backward pass runs -- the native
        backward fires an event per finished arena range (tail convs, then RRDB 22 ... 0, then conv1).
        Default ($RESR_DP_OVERLAP=1 to turn it on): OFF.  The exchange is then issued once, right behind the backward pass,
        and overlaps the NEXT batch's degradation on its side stream only.  Why off: the backward pass runs its dense blocks
        as chained launches that need every CU (conv3x3_ws.h); an RCCL kernel next to them only delays them (tests/
        test_gpu_chain.py holds CUs with a stand-in kernel), but that combination has not run on a multi-GPU box yet, and
        what overlap buys is small -- 67 MB over xGMI is under a millisecond of a 120 ms step."""
        flat = model.flat_parameters()
        self.warm_up(flat.device)
        self.broadcast_(flat)
        if overlap is None:
            overlap = os.environ.get("RESR_DP_OVERLAP", "0") == "1"
        self.overlap = bool(overlap and self.active)
        if self.overlap:
            model.grad_ready_hook = self.all_reduce_ranges_
        else:
            model.grad_hook = self.all_reduce_mean_

    def attach_ema(self, ema: EMA) -> None:
        """The EMA shadow of every rank = rank 0's.  `EMA.register()` runs inside build_model(), i.e. BEFORE attach() broadcasts
        rank 0's weights, and the ranks' generators restart at different seeds (train_realesrnet.seed_rank): without this the
        ranks other than 0 would average from their own random init, validate a different model and disagree on `is_best`."""
        if not self.active:
            return
        flat = getattr(ema, "_flat_shadow", None)
        if flat is not None:
            dist.broadcast(flat, src=0)
        else:
            for k in sorted(ema.shadow):
                dist.broadcast(ema.shadow[k], src=0)

    def all_reduce_ranges_(self, flat: torch.Tensor, ranges, events) -> None:
        """`ranges` are adjacent, descending element ranges of `flat`, events[i] fires (on the producing stream) when
        ranges[i] is final.  Neighbouring ranges are merged into buckets of >= bucket size; each bucket's all-reduce is
        enqueued on the communication stream behind its last event, so it runs under the kernels still producing the next
        ranges; the caller's stream rejoins at the end."""
        if not self.active:
            return
        buckets = self.merge_ranges(ranges)
        if not flat.is_cuda:                       # host tensors (the gloo tests of this logic): no streams, same buckets
            for lo, hi, _ in buckets:
                dist.all_reduce(flat[lo:hi], op=self._op())
            self._finish_mean_(flat)
            return
        main = torch.cuda.current_stream(flat.device)
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream(device=flat.device)
        works = []
        for lo, hi, last in buckets:
            self._comm.wait_event(events[last])
            with torch.cuda.stream(self._comm):
                works.append(dist.all_reduce(flat[lo:hi], op=self._op(), async_op=True))
        for w in works:
            w.wait()
        main.wait_stream(self._comm)
        self._finish_mean_(flat)

    def merge_ranges(self, ranges) -> list:
        """Adjacent, descending (lo, hi) ranges -> buckets (lo, hi, index of the LAST range inside): a bucket closes as soon as
        it holds >= bucket size elements; the tail of the list closes the last one whatever its size."""
        out, hi_open = [], None
        for i, (lo, hi) in enumerate(ranges):
            if hi_open is None:
                hi_open = hi
            elif hi != prev_lo:
                raise ValueError("all_reduce_ranges_: ranges must be adjacent and descending")
            prev_lo = lo
            if hi_open - lo >= self.bucket_elems or i == len(ranges) - 1:
                out.append((lo, hi_open, i))
                hi_open = None
        return out

    def attach_discriminator(self, discriminator: nn.Module) -> None:
        """Discriminator: identical initial weights AND spectral-norm power-iteration vectors `weight_u` / `weight_v`
        (they then stay identical, being deterministic functions of identical weights).  Its gradients are the sum of TWO
        backward passes (train_realesrgan.py:503-516), so they are reduced once, explicitly, by `all_reduce_grads_` after
        the second one -- not from a per-backward hook."""
        flat = discriminator.flat_parameters()
        self.warm_up(flat.device)
        self.broadcast_(flat)
        if self.active:
            if hasattr(discriminator, "flat_uv"):          # the eight (u, v) pairs as one message
                dist.broadcast(discriminator.flat_uv(), src=0)
            else:
                for buf in discriminator.buffers():
                    dist.broadcast(buf, src=0)

    def all_reduce_grads_(self, params: Iterable[nn.Parameter]) -> None:
        """Mean over ranks of the `.grad` of `params`, as ONE bucketed all-reduce: in place when the gradients already
        sit back to back in one arena (the discriminator's backward hands out views of one), else through a flat copy."""
        if not self.active:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        arena = _arena_of(grads)
        if arena is not None:
            self.all_reduce_mean_(arena)
            return
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        self.all_reduce_mean_(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


class RealESRNetStep:
    """One RealESRNet optimisation step: degrade -> G forward -> L1 -> backward -> Adam -> EMA
    (reference train_realesrnet.py:262-394; AMP loss scaling as :97,:383-391)."""

    def __init__(self, model: Generator, ema: Optional[EMA], optimizer: torch.optim.Optimizer,
                 scaler: Optional["torch.amp.GradScaler"] = None,
                 degrade: Optional[Callable[[torch.Tensor], tuple]] = None) -> None:
        self.model, self.ema, self.optimizer, self.scaler, self.degrade = model, ema, optimizer, scaler, degrade
        self.criterion = nn.L1Loss()

    def __call__(self, hr: torch.Tensor, lr: Optional[torch.Tensor] = None) -> torch.Tensor:
        if lr is None:
            lr, hr = self.degrade(hr)                      # train_realesrnet.py:268-377
        self.model.zero_grad(set_to_none=True)             # :380
        sr = self.model(lr)                                # :384
        loss = losses.l1_loss(self.criterion, sr, hr)      # :385 -- value and unit gradient in one launch (csrc/loss.hip)
        if self.scaler is not None:
            self.scaler.scale(loss).backward()             # :388
            self.scaler.step(self.optimizer)               # :390
            self.scaler.update()                           # :391
        else:
            loss.backward()
            self.optimizer.step()
        if self.ema is not None:
            self.ema.update()                              # :394
        return loss.detach()


class RealESRGANStep:
    """One RealESRGAN optimisation step (reference train_realesrgan.py:459-521): generator update with the
    discriminator frozen (pixel L1 on the USM-sharpened sr + adversarial BCE), then two discriminator
    backwards (real, fake) accumulated into one update, one shared GradScaler updated twice, EMA.

    The VGG19 perceptual term (`content_criterion`, optional): the reference wraps it in `torch.Tensor(...)`
    (:477-478), which detaches it -- it is added to the logged g_loss but never back-propagated.  The same
    happens here with the default `ContentLoss(detached=True)`: detached scalars (and, unlike the reference, they
    stay on the device: no 5 D2H syncs).  `ContentLoss(detached=False)` switches the quirk off: the weighted term
    joins g_loss and back-propagates through VGG19 into the generator -- the graph the reference wrote (model.py:311-335)."""

    def __init__(self, generator, discriminator, ema, g_optimizer, d_optimizer, scaler=None, degrade=None,
                 pixel_weight: float = 1.0, adversarial_weight: float = 0.1, content_criterion=None,
                 content_weight=(0.1, 0.1, 1.0, 1.0, 1.0), return_probabilities: bool = False,
                 dp: Optional[DataParallel] = None) -> None:
        from . import imgproc
        self.g, self.d, self.ema = generator, discriminator, ema
        self.dp = dp          # data parallel: the generator reduces from its grad hook, the discriminator right before its step
        self.return_probabilities = return_probabilities
        self.g_opt, self.d_opt, self.scaler, self.degrade = g_optimizer, d_optimizer, scaler, degrade
        self.pixel_weight, self.adversarial_weight = pixel_weight, adversarial_weight
        self.content, self.content_weight = content_criterion, content_weight
        self.pixel = nn.L1Loss()
        self.adv = nn.BCEWithLogitsLoss()
        dev = next(generator.parameters()).device
        self.usm = imgproc.USMSharp(50, 0).to(dev)

    def _d_requires_grad(self, flag: bool) -> None:
        """train_realesrgan.py:465-466 / :491-492, including the one-tensor alias of `Discriminator.flat_parameter()`."""
        for p in self.d.parameters():
            p.requires_grad = flag
        fp = self.d.__dict__.get("_flat_param")
        if fp is not None:
            fp.requires_grad = flag

    def _d_grad_holders(self):
        fp = self.d.__dict__.get("_flat_param")
        return [fp] if fp is not None else list(self.d.parameters())

    def _content_w(self, device) -> torch.Tensor:
        w = getattr(self, "_content_w_dev", None)
        if w is None or w.device != device:
            w = torch.tensor([float(v) for v in self.content_weight], dtype=torch.float32, device=device)
            self._content_w_dev = w
        return w

    def _backward(self, loss):
        if self.scaler is not None:
            self.scaler.scale(loss).backward()
        else:
            loss.backward()

    def _step(self, opt):
        if self.scaler is not None:
            self.scaler.step(opt)
            self.scaler.update()
        else:
            opt.step()

    def __call__(self, hr: torch.Tensor, lr: Optional[torch.Tensor] = None) -> dict:
        if lr is None:
            lr, hr = self.degrade(hr)
        # :460-461 build `real` / `fake` label tensors; here the constant label is an argument of the fused BCE launch
        # (losses.bce_with_logits_const: value + unit gradient in one launch, csrc/loss.hip)
        self._d_requires_grad(False)                                                       # :465-466
        self.g.zero_grad(set_to_none=True)                                                 # :469
        sr = self.g(lr)                                                                    # :474
        sr_usm = self.usm(sr, 0.5, 10)
        pixel_loss = losses.l1_loss(self.pixel, sr_usm, hr, self.pixel_weight, "pixel")    # :475
        content_loss = None
        if self.content is not None:                                                       # :476-477 (detached)
            cl = self.content(sr_usm, hr)
            packed = getattr(self.content, "last_losses", None)     # the five values as one device tensor (detached path)
            if packed is not None and packed.numel() == len(self.content_weight) and cl[0].data_ptr() == packed.data_ptr():
                content_loss = torch.dot(packed, self._content_w(packed.device))           # one launch instead of five mul + four add
            else:
                content_loss = sum(w * c for w, c in zip(self.content_weight, cl))
        adversarial_loss = losses.bce_with_logits_const(self.adv, self.d(sr), 1.0, self.adversarial_weight, "adv")   # :478
        g_loss = pixel_loss + adversarial_loss                                             # :480 (content term detached, see class doc)
        if content_loss is not None and not getattr(self.content, "detached", True):
            g_loss = g_loss + content_loss                                                 # the quirk switched off
        self._backward(g_loss)                                                             # :483
        self._step(self.g_opt)                                                             # :485-486
        self._d_requires_grad(True)                                                        # :491-492
        self.d.zero_grad(set_to_none=True)                                                 # :495
        hr_out = self.d(hr)                                                                # :499
        d_loss_hr = losses.bce_with_logits_const(self.adv, hr_out, 1.0, 1.0, "d_hr")
        self._backward(d_loss_hr)                                                          # :503
        # :507 `sr.detach().clone()`: no clone here.  Discriminator._run_forward ALIASES a contiguous fp32 input (it copies nothing);
        # what makes that safe is that the native forward consumes x into its NHWC workspace before it returns and the backward
        # pass never re-reads x -- and that nothing below mutates `sr` in place.  An in-place op on `sr` added later needs the clone back.
        sr_out = self.d(sr.detach())
        d_loss_sr = losses.bce_with_logits_const(self.adv, sr_out, 0.0, 1.0, "d_sr")
        self._backward(d_loss_sr)                                                          # :513
        if self.dp is not None:                                                            # one exchange for both backwards
            self.dp.all_reduce_grads_(self._d_grad_holders())
        self._step(self.d_opt)                                                             # :515-516
        if self.ema is not None:
            self.ema.update()                                                              # :520
        out = {"pixel_loss": pixel_loss.detach(), "adversarial_loss": adversarial_loss.detach(),
               "d_loss_hr": d_loss_hr.detach(), "d_loss_sr": d_loss_sr.detach()}
        if content_loss is not None:
            out["content_loss"] = content_loss
        if self.return_probabilities:                                                      # :524-525 (logging only)
            out["d_hr_probability"] = torch.sigmoid(hr_out.detach().float().mean())
            out["d_sr_probability"] = torch.sigmoid(sr_out.detach().float().mean())
        return out


class GraphedStep:
    """The fixed-shape part of a train step -- everything AFTER the degradation: generator / discriminator / VGG19 forward and
    backward, the fused losses, GradScaler bookkeeping, both Adam steps, EMA -- captured once per geometry into ONE hipGraph and
    replayed: ~600 dependent launches per RealESRGAN step leave the host in one call.  The degradation keeps running eagerly on its
    side stream (its plan changes per batch); its LR / HR outputs are copied into the graph's static inputs.

    Requirements (checked): optimisers built with `capturable=True` (the step counter lives on the device) and, next to a
    GradScaler, `fused=True` (the scaler's step of a non-fused optimiser reads found_inf on the host: a sync inside the capture);
    no data-parallel exchange inside the step (`dp` inactive); the same shapes on every call.  The first `warmup` calls
    run eagerly (one-time allocations, kernel attributes, optimizer state); a change of shape or of a FLOAT learning rate
    re-captures -- a scheduler that moves the rate every iteration would therefore never let a graph replay (a warning says so):
    give such optimisers a TENSOR learning rate (`lr=torch.tensor(2e-4, device=...)`: capturable Adam reads it on the device, the
    scheduler updates it in place, and it is not part of the capture key).  Returns what the wrapped step returns -- tensors that
    the NEXT replay overwrites.
    Chained dense-block launches are captured like any other (csrc/conv3x3_ws.hip): do not replay while ANOTHER stream runs
    chained launches on the same device (include/resr.h)."""

    def __init__(self, step, warmup: int = 3) -> None:
        self.step, self.warmup = step, max(1, int(warmup))
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._key = None
        self._calls = 0
        self._hr = self._lr = self._out = None
        self._recaptures = 0
        for opt in self._optimizers():
            if not opt.defaults.get("capturable", False):
                raise ValueError("GraphedStep: build the optimisers with capturable=True (their step counters must live on the device)")
            if getattr(step, "scaler", None) is not None and not opt.defaults.get("fused", False):
                raise ValueError("GraphedStep: next to a GradScaler the optimisers must be fused=True (scaler.step() of a non-fused "
                                 "optimiser calls .item() on found_inf, which aborts a stream capture)")
        dp = getattr(step, "dp", None)
        if dp is not None and getattr(dp, "active", False):
            raise ValueError("GraphedStep: a data-parallel exchange inside the step is not captured; run the step eagerly")

    def _optimizers(self):
        return [o for o in (getattr(self.step, n, None) for n in ("optimizer", "g_opt", "d_opt")) if o is not None]

    def _lrs(self):
        # float rates are baked into the captured launches: part of the key.  Tensor rates live on the device (capturable Adam
        # reads them there) and are updated in place: not part of the key, and never read back (float() would be a sync per call)
        return tuple(float(g["lr"]) for o in self._optimizers() for g in o.param_groups if not torch.is_tensor(g["lr"]))

    def __call__(self, hr: torch.Tensor, lr: Optional[torch.Tensor] = None):
        if lr is None:
            lr, hr = self.step.degrade(hr)
        key = (tuple(hr.shape), tuple(lr.shape), str(hr.device), self._lrs())
        if self._graph is None or key != self._key:
            if key != self._key:
                if self._key is not None:
                    self._recaptures += 1
                    if self._recaptures in (8, 64, 512):
                        import warnings
                        warnings.warn(f"GraphedStep: capture key changed {self._recaptures} times (shape or a float learning rate): the step "
                                      "keeps running eagerly; use tensor learning rates with a per-iteration scheduler", RuntimeWarning)
                self._graph, self._calls, self._key = None, 0, key
            if self._calls < self.warmup:
                self._calls += 1
                return self.step(hr, lr)
            self._hr, self._lr = hr.clone(), lr.clone()
            torch.cuda.synchronize(hr.device)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    self._out = self.step(self._hr, self._lr)
            except Exception:
                self._graph, self._key, self._calls = None, None, 0      # a failed capture leaves nothing half-set
                raise
            self._graph = g
            g.replay()                  # the capture itself executes nothing: this call's step is the first replay
            return self._out
        self._hr.copy_(hr)
        self._lr.copy_(lr)
        self._graph.replay()
        return self._out
