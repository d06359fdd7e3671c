"""RealESRNet training entry point behind the reference's `train_realesrnet.py` surface: same
function names, epoch loop, checkpoint dictionary and file names, NIQE validation with the EMA weights applied.

    python -m real_esrgan_pytorch_amd.train_realesrnet          # reads real_esrgan_pytorch_amd.config

The step itself is `train.RealESRNetStep` (degradation -> generator forward/backward on the HIP kernels -> Adam ->
EMA); the degradation uses the blur kernels the dataset sampled for each image (reference train_realesrnet.py:258-413).
TensorBoard is absent from the image: scalars go to `samples/logs/<exp>/scalars.jsonl` through the same `add_scalar`.
"""
from __future__ import annotations

import json
import os
import shutil
import time
from typing import Any, List, Optional

import numpy as np
import torch
from torch import nn, optim
from torch.optim import lr_scheduler
from torch.utils.data import DataLoader

from . import config, imgproc
from . import _lib
from .dataset import CUDAPrefetcher, TestImageDataset, TrainValidImageDataset
from .degrade import DegradationPrefetcher
from .image_quality_assessment import NIQE
from .meters import AverageMeter, ProgressMeter, Summary  # noqa: F401  (the reference's script-level names)
from .model import EMA, Generator
from .train import DataParallel, RealESRNetStep, setup_distributed


class ScalarWriter:
    """`SummaryWriter.add_scalar` stand-in: one JSON line per scalar (rank 0 only under data parallelism)."""

    def __init__(self, log_dir: str, enabled: bool = True) -> None:
        self.enabled = enabled
        if enabled:
            os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")

    def add_scalar(self, tag: str, value: float, step: int) -> None:
        if not self.enabled:
            return
        with open(self.path, "a") as f:
            f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")


_RANK, _WORLD = 0, 1     # set by main() through train.setup_distributed()


def seed_rank(rank: int, world: int, base: int = 0) -> None:
    """Data-parallel ranks must not draw the same degradation plans, sigma / quality values and crop offsets: the host and
    device generators of rank r restart at base + r (rank 0 keeps the reference's seeds, config.py:64-66)."""
    if world > 1:
        import random
        random.seed(base + rank)
        np.random.seed(base + rank)
        torch.manual_seed(base + rank)


def load_dataset() -> List[CUDAPrefetcher]:
    """Reference train_realesrnet.py:131-172."""
    train_datasets = TrainValidImageDataset(config.train_image_dir, config.image_size, config.upscale_factor, "Train",
                                            config.degradation_model_parameters_dict)
    valid_datasets = TrainValidImageDataset(config.valid_image_dir, config.image_size, config.upscale_factor, "Valid",
                                            config.degradation_model_parameters_dict)
    test_datasets = TestImageDataset(config.test_lr_image_dir, config.test_hr_image_dir)
    workers = config.num_workers
    # data parallel: every rank draws its own shard of each epoch (batch_size is per GPU, as in the reference's per-process config)
    sampler = torch.utils.data.distributed.DistributedSampler(train_datasets, _WORLD, _RANK, shuffle=True) if _WORLD > 1 else None
    train_dataloader = DataLoader(train_datasets, batch_size=config.batch_size, shuffle=sampler is None, sampler=sampler,
                                  num_workers=workers, pin_memory=True, drop_last=True, persistent_workers=workers > 0)
    valid_dataloader = DataLoader(valid_datasets, batch_size=1, shuffle=False, num_workers=min(1, workers),
                                  pin_memory=True, drop_last=False, persistent_workers=workers > 0)
    test_dataloader = DataLoader(test_datasets, batch_size=1, shuffle=False, num_workers=min(1, workers),
                                 pin_memory=True, drop_last=False, persistent_workers=workers > 0)
    return [CUDAPrefetcher(train_dataloader, config.device), CUDAPrefetcher(valid_dataloader, config.device),
            CUDAPrefetcher(test_dataloader, config.device)]


def build_model() -> List[nn.Module]:
    """Reference train_realesrnet.py:175-183."""
    model = Generator(config.in_channels, config.out_channels, config.upscale_factor,
                      precision=getattr(config, "precision", "fast")).to(device=config.device)
    ema_model = EMA(model, config.ema_model_weight_decay).to(device=config.device)
    ema_model.register()
    return [model, ema_model]


def define_loss() -> nn.L1Loss:
    return nn.L1Loss().to(device=config.device)


def define_optimizer(model) -> optim.Adam:
    return optim.Adam(model.parameters(), config.model_lr, config.model_betas)


def define_scheduler(optimizer) -> lr_scheduler.StepLR:
    return lr_scheduler.StepLR(optimizer, config.lr_scheduler_step_size, config.lr_scheduler_gamma)


def load_checkpoint(path: str, model: nn.Module, ema_model: nn.Module, optimizer, scheduler) -> List[Any]:
    """Reference train_realesrnet.py:60-83: keys present in the current modules are taken, the rest ignored."""
    checkpoint = torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
    for module, key in ((model, "state_dict"), (ema_model, "ema_state_dict")):
        current = module.state_dict()
        current.update({k: v for k, v in checkpoint[key].items() if k in current})
        module.load_state_dict(current)
    optimizer.load_state_dict(checkpoint["optimizer"])
    scheduler.load_state_dict(checkpoint["scheduler"])
    return [checkpoint["epoch"], checkpoint["best_niqe"]]


def save_checkpoint(epoch: int, best_niqe: float, is_best: bool, model, ema_model, optimizer, scheduler,
                    samples_dir: str, results_dir: str) -> str:
    """Reference train_realesrnet.py:117-129: same dictionary, same file names."""
    path = os.path.join(samples_dir, f"g_epoch_{epoch + 1}.pth.tar")
    torch.save({"epoch": epoch + 1, "best_niqe": best_niqe, "state_dict": model.state_dict(),
                "ema_state_dict": ema_model.state_dict(), "optimizer": optimizer.state_dict(),
                "scheduler": scheduler.state_dict()}, path)
    if is_best:
        shutil.copyfile(path, os.path.join(results_dir, "g_best.pth.tar"))
    if (epoch + 1) == config.epochs:
        shutil.copyfile(path, os.path.join(results_dir, "g_last.pth.tar"))
    return path


def main() -> None:
    global _RANK, _WORLD
    _RANK, _WORLD, device = setup_distributed()          # one process per GPU: cuda:LOCAL_RANK, RCCL group when WORLD_SIZE > 1
    config.device = device
    seed_rank(_RANK, _WORLD)
    start_epoch, best_niqe = 0, 100.0
    train_prefetcher, valid_prefetcher, test_prefetcher = load_dataset()
    model, ema_model = build_model()
    dp = DataParallel()
    dp.attach(model)                                     # broadcast of the initial weights + all-reduce of the gradient arena per step
    dp.attach_ema(ema_model)                             # ... and of the EMA shadow (registered from each rank's own init above)
    pixel_criterion = define_loss()
    optimizer = define_optimizer(model)
    scheduler = define_scheduler(optimizer)
    if config.resume:
        start_epoch, best_niqe = load_checkpoint(config.resume, model, ema_model, optimizer, scheduler)
        print("Loaded pretrained model weights.")
        dp.broadcast_(model.flat_parameters())           # every rank read the same file; keep the replicas bit-identical anyway
        dp.attach_ema(ema_model)
    samples_dir = os.path.join("samples", config.exp_name)
    results_dir = os.path.join("results", config.exp_name)
    os.makedirs(samples_dir, exist_ok=True)
    os.makedirs(results_dir, exist_ok=True)
    writer = ScalarWriter(os.path.join("samples", "logs", config.exp_name), enabled=_RANK == 0)
    scaler = torch.amp.GradScaler("cuda") if getattr(config, "precision", "fast") != "strict" else None
    niqe_model = NIQE(config.upscale_factor, config.niqe_model_path).to(device=config.device)
    for epoch in range(start_epoch, config.epochs):
        sampler = getattr(train_prefetcher.original_dataloader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        train(model, ema_model, train_prefetcher, pixel_criterion, optimizer, epoch, scaler, writer)
        _lib.chain_health(sync=True)   # fail loudly if a chained conv launch ever gave up on a neighbouring tile
        _ = validate(model, ema_model, valid_prefetcher, epoch, writer, niqe_model, "Valid")
        niqe = validate(model, ema_model, test_prefetcher, epoch, writer, niqe_model, "Test")
        print("\n")
        scheduler.step()
        is_best = niqe < best_niqe
        best_niqe = min(niqe, best_niqe)
        if _RANK == 0:       # weights, EMA shadow and optimiser state are identical on every rank: one writer
            save_checkpoint(epoch, best_niqe, is_best, model, ema_model, optimizer, scheduler, samples_dir, results_dir)


def train(model: nn.Module, ema_model: nn.Module, train_prefetcher: CUDAPrefetcher, pixel_criterion: nn.L1Loss,
          optimizer: optim.Adam, epoch: int, scaler: Optional["torch.amp.GradScaler"], writer: ScalarWriter) -> None:
    """Reference train_realesrnet.py:218-413."""
    jpeg_operation = imgproc.DiffJPEG(False)
    usm_sharpener = imgproc.USMSharp(50, 0).to(device=config.device)
    batches = len(train_prefetcher)
    batch_time = AverageMeter("Time", ":6.3f", Summary.NONE)            # :219-224
    data_time = AverageMeter("Data", ":6.3f", Summary.NONE)
    losses = AverageMeter("Loss", ":6.6f", Summary.NONE)
    progress = ProgressMeter(batches, [batch_time, data_time, losses], prefix=f"Epoch: [{epoch + 1}]")
    model.train()

    # The second-order degradation (:262-377: host draws in the reference's order, blur kernels from the dataset batch) runs
    # ONE BATCH AHEAD on a side stream: batch i+1's kernels are enqueued before step i is issued and run under it.
    degraded = DegradationPrefetcher(train_prefetcher, usm_sharpener, jpeg_operation, config.upscale_factor, config.image_size,
                                     config.device)
    step = RealESRNetStep(model, ema_model, optimizer, scaler, None)
    step.criterion = pixel_criterion
    batch_index = 0
    degraded.reset()
    item = degraded.next()
    end = time.time()
    while item is not None:
        data_time.update(time.time() - end)
        lr, hr, _ = item
        loss = step(hr, lr)
        losses.update(loss, hr.size(0))                    # :397 -- every batch counts; accumulated on the device, no read-back
        batch_time.update(time.time() - end)
        if batch_index % config.print_frequency == 0:      # the only host read-back of the loop
            writer.add_scalar("Train/Loss", losses.val, batch_index + epoch * batches + 1)
            _lib.chain_health()                            # two host-mapped counters, no synchronisation: a broken chained launch aborts here
            if _RANK == 0:
                progress.display(batch_index)
        end = time.time()
        item = degraded.next()
        batch_index += 1


def validate(model: nn.Module, ema_model: nn.Module, data_prefetcher: CUDAPrefetcher, epoch: int, writer: ScalarWriter,
             niqe_model: Any, mode: str) -> float:
    """Reference train_realesrnet.py:416-488: EMA weights applied for the evaluation, restored afterwards."""
    if mode not in ("Valid", "Test"):
        raise ValueError("Unsupported mode, please use `Valid` or `Test`.")
    batch_time = AverageMeter("Time", ":6.3f", Summary.NONE)
    niqe_metrics = AverageMeter("NIQE", ":4.2f", Summary.AVERAGE)
    ema_model.apply_shadow()
    model.eval()
    batch_index = 0
    data_prefetcher.reset()
    batch_data = data_prefetcher.next()
    end = time.time()
    with torch.no_grad():
        while batch_data is not None:
            lr = batch_data["lr"].to(device=config.device, non_blocking=True)
            sr = model(lr)
            niqe = niqe_model(sr)
            niqe_metrics.update(niqe.reshape(()), lr.size(0))
            batch_time.update(time.time() - end)
            end = time.time()
            batch_data = data_prefetcher.next()
            batch_index += 1
    ema_model.restore()
    avg_niqe = niqe_metrics.avg     # the meters accumulate on the device: THIS read-back is the evaluation's first synchronisation
    _lib.chain_health()             # ... after which every launch of the evaluation has reported
    if _RANK == 0:
        ProgressMeter(len(data_prefetcher), [batch_time, niqe_metrics], prefix=f"{mode}: ").display_summary()
    writer.add_scalar(f"{mode}/NIQE", avg_niqe, epoch + 1)
    return avg_niqe


if __name__ == "__main__":
    main()
