"""Directory evaluation behind the reference's `test.py` (reference test.py:29-96): super-resolve every image of
`config.lr_dir` with the EMA weights of `config.model_path`, write the results to `config.sr_dir`, report the mean NIQE.

    RESR_MODE=test python -m real_esrgan_pytorch_amd.test

PIL does the file I/O (cv2 is absent from the image; the reference's BGR<->RGB swaps cancel out) and a local natural
sort replaces `natsort`.
"""
from __future__ import annotations

import os
import re
from typing import List

import torch

from . import _lib, config, imgproc
from .image_quality_assessment import NIQE
from .model import Generator
from .tiling import super_resolve


def natural_sorted(names: List[str]) -> List[str]:
    """`natsort.natsorted` for plain file names: digit runs compare as integers."""
    return sorted(names, key=lambda s: [(0, int(t), "") if t.isdigit() else (1, 0, t) for t in re.split(r"(\d+)", s) if t])


def main() -> float:
    torch.cuda.set_device(config.device)            # one process per GPU: every launch and side stream on this device
    from PIL import Image
    model = Generator(config.in_channels, config.out_channels, config.upscale_factor,
                      precision=config.inference_precision)                     # fp32 call site: test.py:79-80 (no autocast)
    model = model.to(device=config.device, memory_format=torch.channels_last)              # test.py:32
    print("Build Real_ESRGAN model successfully.")
    checkpoint = torch.load(config.model_path, map_location=lambda storage, loc: storage, weights_only=False)
    current = model.state_dict()
    model.load_state_dict({k.replace("model.", ""): v for k, v in checkpoint["ema_state_dict"].items()
                           if k.replace("model.", "") in current})                         # test.py:36-40
    print(f"Load Real_ESRGAN model weights `{os.path.abspath(config.model_path)}` successfully.")
    os.makedirs(config.sr_dir, exist_ok=True)
    model.eval()
    niqe = NIQE(config.upscale_factor, config.niqe_model_path).to(device=config.device)
    niqe_metrics = 0.0
    file_names = natural_sorted(os.listdir(config.lr_dir))
    total_files = len(file_names)
    for name in file_names:
        lr_image_path = os.path.join(config.lr_dir, name)
        print(f"Processing `{os.path.abspath(lr_image_path)}`...")
        lr_tensor = imgproc.image_to_tensor(imgproc.read_image_rgb(lr_image_path), False, False).unsqueeze_(0)
        lr_tensor = lr_tensor.to(device=config.device, memory_format=torch.channels_last, non_blocking=True)
        with torch.no_grad():
            sr_tensor = super_resolve(model, lr_tensor)                                    # test.py:79 (any frame size)
        Image.fromarray(imgproc.tensor_to_image(sr_tensor, False, False)).save(os.path.join(config.sr_dir, name))
        niqe_metrics += niqe(sr_tensor).item()
    _lib.chain_health()             # fail loudly if a chained conv launch ever gave up on a neighbouring tile
    avg_niqe = 100 if niqe_metrics / total_files > 100 else niqe_metrics / total_files    # test.py:92
    print(f"NIQE: {avg_niqe:4.2f} 100u")
    return avg_niqe


if __name__ == "__main__":
    main()
