"""RealESRGAN training entry point behind the reference's `train_realesrgan.py` surface: same
function names, epoch loop, the two checkpoint dictionaries and their file names (`d_epoch_N` / `g_epoch_N`,
`d_best` / `g_best`, `d_last` / `g_last`), the three ways of resuming, NIQE validation under the EMA weights.

    RESR_MODE=train_realesrgan python -m real_esrgan_pytorch_amd.train_realesrgan

The step is `train.RealESRGANStep` (degradation -> generator update with the discriminator frozen -> two discriminator
backwards -> EMA; reference train_realesrgan.py:459-521) on the HIP kernels; the degradation uses the blur kernels the
dataset sampled for each image, USM-sharpened ground truth included (reference :330-452).  Scalars go to
`samples/logs/<exp>/scalars.jsonl` under the reference's tags.
"""
from __future__ import annotations

import os
import shutil
import time
from typing import Any, List, Optional

import torch
from torch import nn, optim
from torch.optim import lr_scheduler

from . import config, imgproc
from . import _lib
from .content_loss import ContentLoss
from .dataset import CUDAPrefetcher
from .degrade import DegradationPrefetcher
from .discriminator import Discriminator
from .image_quality_assessment import NIQE
from .model import EMA, Generator
from . import train_realesrnet as _net
from .train import DataParallel, RealESRGANStep, setup_distributed
from .meters import AverageMeter, ProgressMeter, Summary  # noqa: F401
from .train_realesrnet import ScalarWriter, load_dataset, validate  # noqa: F401


def build_model() -> List[nn.Module]:
    """Reference train_realesrgan.py:223-237."""
    precision = getattr(config, "precision", "fast")
    discriminator = Discriminator(precision=precision).to(device=config.device)
    generator = Generator(config.in_channels, config.out_channels, config.upscale_factor,
                          precision=precision).to(device=config.device)
    ema_model = EMA(generator, config.ema_model_weight_decay).to(device=config.device)
    ema_model.register()
    return [discriminator, generator, ema_model]


def define_loss() -> List[nn.Module]:
    """Reference train_realesrgan.py:240-253.  The VGG19 weights are torchvision's in the reference (a download);
    here they are whatever `content_criterion.load_state_dict` is given, random otherwise -- the term is logged only
    (see `train.RealESRGANStep`)."""
    pixel_criterion = nn.L1Loss().to(device=config.device)
    content_criterion = ContentLoss(config.feature_model_extractor_nodes, config.feature_model_normalize_mean,
                                    config.feature_model_normalize_std,
                                    precision=getattr(config, "precision", "fast")).to(device=config.device)
    adversarial_criterion = nn.BCEWithLogitsLoss().to(device=config.device)
    return [pixel_criterion, content_criterion, adversarial_criterion]


def define_optimizer(discriminator: nn.Module, generator: nn.Module) -> List[optim.Adam]:
    return [optim.Adam(discriminator.parameters(), config.model_lr, config.model_betas),
            optim.Adam(generator.parameters(), config.model_lr, config.model_betas)]


def define_scheduler(d_optimizer: optim.Adam, g_optimizer: optim.Adam) -> List[lr_scheduler.MultiStepLR]:
    return [lr_scheduler.MultiStepLR(d_optimizer, config.lr_scheduler_milestones, config.lr_scheduler_gamma),
            lr_scheduler.MultiStepLR(g_optimizer, config.lr_scheduler_milestones, config.lr_scheduler_gamma)]


def _load(path: str) -> dict:
    return torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)


def _update_state(module: nn.Module, saved: dict) -> None:
    """Keys present in the current module are taken, the rest ignored (reference :75-79, :93-104)."""
    current = module.state_dict()
    current.update({k: v for k, v in saved.items() if k in current})
    module.load_state_dict(current)


_DP: Optional[DataParallel] = None     # set by main(); `train()` hands it to the step


def main() -> None:
    global _DP
    rank, world, device = setup_distributed()            # one process per GPU: cuda:LOCAL_RANK, RCCL group when WORLD_SIZE > 1
    _net._RANK, _net._WORLD = rank, world
    config.device = device
    _net.seed_rank(rank, world)
    start_epoch, best_niqe = 0, 100.0
    train_prefetcher, valid_prefetcher, test_prefetcher = load_dataset()
    discriminator, generator, ema_model = build_model()
    pixel_criterion, content_criterion, adversarial_criterion = define_loss()
    d_optimizer, g_optimizer = define_optimizer(discriminator, generator)
    d_scheduler, g_scheduler = define_scheduler(d_optimizer, g_optimizer)
    if config.resume:                                     # RealESRNet weights into the generator (:60-65)
        generator.load_state_dict(_load(config.resume)["state_dict"])
        print("Loaded RealESRNet model weights.")
    if getattr(config, "resume_d", ""):                   # :68-84
        checkpoint = _load(config.resume_d)
        start_epoch, best_niqe = checkpoint["epoch"], checkpoint["best_niqe"]
        _update_state(discriminator, checkpoint["state_dict"])
        d_optimizer.load_state_dict(checkpoint["optimizer"])
        d_scheduler.load_state_dict(checkpoint["scheduler"])
        print("Loaded pretrained discriminator model weights.")
    if getattr(config, "resume_g", ""):                   # :87-110
        checkpoint = _load(config.resume_g)
        start_epoch, best_niqe = checkpoint["epoch"], checkpoint["best_niqe"]
        _update_state(generator, checkpoint["state_dict"])
        _update_state(ema_model, checkpoint["ema_state_dict"])
        g_optimizer.load_state_dict(checkpoint["optimizer"])
        g_scheduler.load_state_dict(checkpoint["scheduler"])
        print("Loaded pretrained generator model weights.")
    # data parallel (after every way of loading weights): rank 0's generator, discriminator and spectral-norm u / v everywhere;
    # generator gradients reduced from its backward hook, discriminator gradients once after its second backward
    _DP = DataParallel()
    _DP.attach(generator)
    _DP.attach_ema(ema_model)                             # registered from each rank's own init in build_model(): rank 0's everywhere
    _DP.attach_discriminator(discriminator)
    samples_dir = os.path.join("samples", config.exp_name)
    results_dir = os.path.join("results", config.exp_name)
    os.makedirs(samples_dir, exist_ok=True)
    os.makedirs(results_dir, exist_ok=True)
    writer = ScalarWriter(os.path.join("samples", "logs", config.exp_name), enabled=rank == 0)
    scaler = torch.amp.GradScaler("cuda") if getattr(config, "precision", "fast") != "strict" else None
    niqe_model = NIQE(config.upscale_factor, config.niqe_model_path).to(device=config.device)
    for epoch in range(start_epoch, config.epochs):
        sampler = getattr(train_prefetcher.original_dataloader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        train(discriminator, generator, ema_model, train_prefetcher, pixel_criterion, content_criterion,
              adversarial_criterion, d_optimizer, g_optimizer, epoch, scaler, writer)
        _lib.chain_health(sync=True)   # fail loudly if a chained conv launch ever gave up on a neighbouring tile
        _ = validate(generator, ema_model, valid_prefetcher, epoch, writer, niqe_model, "Valid")
        niqe = validate(generator, ema_model, test_prefetcher, epoch, writer, niqe_model, "Test")
        print("\n")
        d_scheduler.step()
        g_scheduler.step()
        is_best = niqe < best_niqe
        best_niqe = min(niqe, best_niqe)
        if rank != 0:        # weights, EMA shadow and optimiser states are identical on every rank: one writer
            continue
        d_path = os.path.join(samples_dir, f"d_epoch_{epoch + 1}.pth.tar")
        g_path = os.path.join(samples_dir, f"g_epoch_{epoch + 1}.pth.tar")
        torch.save({"epoch": epoch + 1, "best_niqe": best_niqe, "state_dict": discriminator.state_dict(),
                    "optimizer": d_optimizer.state_dict(), "scheduler": d_scheduler.state_dict()}, d_path)
        torch.save({"epoch": epoch + 1, "best_niqe": best_niqe, "state_dict": generator.state_dict(),
                    "ema_state_dict": ema_model.state_dict(), "optimizer": g_optimizer.state_dict(),
                    "scheduler": g_scheduler.state_dict()}, g_path)
        if is_best:
            shutil.copyfile(d_path, os.path.join(results_dir, "d_best.pth.tar"))
            shutil.copyfile(g_path, os.path.join(results_dir, "g_best.pth.tar"))
        if (epoch + 1) == config.epochs:
            shutil.copyfile(d_path, os.path.join(results_dir, "d_last.pth.tar"))
            shutil.copyfile(g_path, os.path.join(results_dir, "g_last.pth.tar"))


def train(discriminator: nn.Module, generator: nn.Module, ema_model: nn.Module, train_prefetcher: CUDAPrefetcher,
          pixel_criterion: nn.L1Loss, content_criterion: Optional[ContentLoss],
          adversarial_criterion: nn.BCEWithLogitsLoss, d_optimizer: optim.Adam, g_optimizer: optim.Adam, epoch: int,
          scaler: Optional["torch.amp.GradScaler"], writer: Any) -> None:
    """Reference train_realesrgan.py:282-553."""
    jpeg_operation = imgproc.DiffJPEG(False)
    usm_sharpener = imgproc.USMSharp(50, 0).to(device=config.device)
    batches = len(train_prefetcher)
    names = ["pixel_loss", "content_loss", "adversarial_loss", "d_loss_hr", "d_loss_sr", "d_hr_probability", "d_sr_probability"]
    batch_time = AverageMeter("Time", ":6.3f", Summary.NONE)              # :295-309
    meters = {k: AverageMeter(label, fmt, Summary.NONE) for k, label, fmt in (
        ("pixel_loss", "Pixel loss", ":6.6f"), ("content_loss", "Content loss", ":6.6f"), ("adversarial_loss", "Adversarial loss", ":6.6f"),
        ("d_loss_hr", "D(HR) loss", ":6.6f"), ("d_loss_sr", "D(SR) loss", ":6.6f"),
        ("d_hr_probability", "D(HR)", ":6.3f"), ("d_sr_probability", "D(SR)", ":6.3f"))}
    progress = ProgressMeter(batches, [batch_time] + [meters[k] for k in names], prefix=f"Epoch: [{epoch + 1}]")
    discriminator.train()
    generator.train()

    # the second-order degradation (:330-452) one batch ahead on a side stream, under the previous batch's step
    degraded = DegradationPrefetcher(train_prefetcher, usm_sharpener, jpeg_operation, config.upscale_factor, config.image_size,
                                     config.device)
    step = RealESRGANStep(generator, discriminator, ema_model, g_optimizer, d_optimizer, scaler, None,
                          config.pixel_weight, config.adversarial_weight, content_criterion, config.content_weight,
                          return_probabilities=True, dp=_DP)
    step.pixel, step.adv = pixel_criterion, adversarial_criterion
    batch_index = 0
    degraded.reset()
    item = degraded.next()
    end = time.time()
    while item is not None:
        lr, hr, _ = item
        out = step(hr, lr)
        zero = out["pixel_loss"].new_zeros(())
        for k in names:                                    # :527-535 -- every batch counts; accumulated on the device, no read-back
            meters[k].update(out.get(k, zero).float().reshape(()), hr.size(0))
        batch_time.update(time.time() - end)
        if batch_index % config.print_frequency == 0:
            _lib.chain_health()      # two host-mapped counters, no synchronisation
            # one D2H transfer for everything the log needs (the reference does 5-12 `.item()` syncs per step)
            vals = dict(zip(names, torch.stack([out.get(k, zero).float().reshape(()) for k in names]).tolist()))
            iters = batch_index + epoch * batches + 1
            writer.add_scalar("Train/D_Loss", vals["d_loss_hr"] + vals["d_loss_sr"], iters)
            writer.add_scalar("Train/G_Loss", vals["pixel_loss"] + vals["content_loss"] + vals["adversarial_loss"], iters)
            writer.add_scalar("Train/Pixel_Loss", vals["pixel_loss"], iters)
            writer.add_scalar("Train/Content_Loss", vals["content_loss"], iters)
            writer.add_scalar("Train/Adversarial_Loss", vals["adversarial_loss"], iters)
            writer.add_scalar("Train/D(HR)_Probability", vals["d_hr_probability"], iters)
            writer.add_scalar("Train/D(SR)_Probability", vals["d_sr_probability"], iters)
            if _net._RANK == 0:
                progress.display(batch_index)
        end = time.time()
        item = degraded.next()
        batch_index += 1


if __name__ == "__main__":
    main()
