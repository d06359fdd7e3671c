"""Single-image super-resolution entry point with the reference's CLI (reference inference.py:25-70):

    python -m real_esrgan_pytorch_amd.inference --inputs_path lr.png --output_path sr.png --weights_path g.pth.tar

Same flow as the reference: build `Generator`, load `checkpoint["state_dict"]` with the "model." prefix
stripped, read the image as RGB float in [0,1], run the model under no_grad on the whole image (frames beyond the conv
kernels' 2^24-pixel tensors -- a 1080p LR input of the x4 model -- are cut into haloed tiles and stitched: tiling.super_resolve), write
`tensor_to_image` (truncating uint8 conversion, imgproc.py:1594).  Image I/O uses PIL (cv2 is not part of
this environment); the BGR<->RGB swaps of the reference cancel out and are therefore absent.
"""
import argparse

import numpy as np
import torch

from . import _lib, config, imgproc
from .model import Generator
from .tiling import super_resolve


def main(args) -> None:
    from PIL import Image
    torch.cuda.set_device(config.device)            # one process per GPU: every launch and side stream on this device
    model = Generator(config.in_channels, config.out_channels, config.upscale_factor,
                      precision=getattr(args, "precision", None) or config.inference_precision)   # fp32 call site: inference.py:52-53
    model = model.to(memory_format=torch.channels_last, device=config.device)        # inference.py:28
    print("Build Real_ESRGAN model successfully.")
    checkpoint = torch.load(args.weights_path, map_location=lambda storage, loc: storage, weights_only=False)
    model.load_state_dict({k.replace("model.", ""): v for k, v in checkpoint["state_dict"].items()})   # inference.py:33
    print(f"Load Real_ESRGAN model weights `{args.weights_path}` successfully.")
    model.eval()
    lr_image = np.asarray(Image.open(args.inputs_path).convert("RGB")).astype(np.float32) / 255.0
    lr_tensor = imgproc.image_to_tensor(lr_image, False, False).unsqueeze_(0)
    lr_tensor = lr_tensor.to(device=config.device, memory_format=torch.channels_last, non_blocking=True)
    with torch.no_grad():
        sr_tensor = super_resolve(model, lr_tensor)                                    # inference.py:53 (any frame size)
    sr_image = imgproc.tensor_to_image(sr_tensor, False, False)
    _lib.chain_health()             # (the image is on the host: every launch has reported) a broken chained launch must not reach the file
    Image.fromarray(sr_image).save(args.output_path)
    print(f"SR image save to `{args.output_path}`")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Using the Real_ESRGAN model generator super-resolution images.")
    parser.add_argument("--inputs_path", type=str, help="Low-resolution image path.")
    parser.add_argument("--output_path", type=str, help="Super-resolution image path.")
    parser.add_argument("--weights_path", type=str, help="Model weights file path.")
    parser.add_argument("--precision", type=str, default=None, choices=["fast", "exact16", "strict"],
                        help="kernel arithmetic; default config.inference_precision = exact16 (the reference runs fp32 here)")
    main(parser.parse_args())
