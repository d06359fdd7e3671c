"""Host side of the data path behind the reference's `dataset.py` surface: file reading,
train-time augmentation, per-sample blur / sinc kernel sampling, validation LR synthesis and the HBM batch stager.

Same class names, constructor arguments and batch dictionaries as the reference (dataset.py:27-30); images are read
with PIL instead of cv2 (absent from the image).  All of it is CPU work done in DataLoader workers, exactly as in the
reference -- the device side of a batch starts in `degrade.Degrader` / `train.RealESRNetStep`.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import imgproc
from .degrade import sample_blur_kernels

__all__ = ["TrainValidImageDataset", "TestImageDataset", "PrefetchGenerator", "PrefetchDataLoader", "CPUPrefetcher", "CUDAPrefetcher"]


class TrainValidImageDataset(Dataset):
    """Reference dataset.py:33-162.  "Train": rotate / flip augmentation of the full-size HR image + the three blur
    kernels of the sample; "Valid": centre crop + MATLAB-bicubic LR."""

    def __init__(self, image_dir: str, image_size: int, upscale_factor: int, mode: str,
                 degradation_model_parameters_dict: dict) -> None:
        super().__init__()
        self.image_file_names = [os.path.join(image_dir, name) for name in os.listdir(image_dir)]
        self.image_size = image_size
        self.parameters = degradation_model_parameters_dict
        self.upscale_factor = upscale_factor
        self.mode = mode

    def __getitem__(self, batch_index: int) -> dict:
        image = imgproc.read_image_rgb(self.image_file_names[batch_index])           # dataset.py:66 (+ :75 BGR->RGB)
        if self.mode == "Train":
            hr_image = imgproc.random_rotate(image, [0, 90, 180, 270])               # :70
            hr_image = imgproc.random_horizontally_flip(hr_image, 0.5)               # :71
            hr_image = imgproc.random_vertically_flip(hr_image, 0.5)                 # :72
            hr_tensor = imgproc.image_to_tensor(hr_image, False, False)              # :79
            kernel1, kernel2, sinc_kernel = sample_blur_kernels(self.parameters)     # :82-137, same draw order
            return {"hr": hr_tensor, "kernel1": torch.from_numpy(kernel1), "kernel2": torch.from_numpy(kernel2),
                    "sinc_kernel": torch.from_numpy(sinc_kernel)}
        if self.mode == "Valid":
            hr_image = imgproc.center_crop(image, self.image_size)                   # :146
            lr_image = imgproc.image_resize(hr_image, 1 / self.upscale_factor)       # :148
            return {"lr": imgproc.image_to_tensor(lr_image, False, False),
                    "hr": imgproc.image_to_tensor(np.ascontiguousarray(hr_image), False, False)}
        raise ValueError("Unsupported data processing model, please use `Train` or `Valid`.")

    def __len__(self) -> int:
        return len(self.image_file_names)


class TestImageDataset(Dataset):
    """Reference dataset.py:165-197 (paired LR / HR folders; HR names follow the LR listing, as there)."""

    def __init__(self, test_lr_image_dir: str, test_hr_image_dir: str) -> None:
        super().__init__()
        names = os.listdir(test_lr_image_dir)
        self.lr_image_file_names = [os.path.join(test_lr_image_dir, x) for x in names]
        self.hr_image_file_names = [os.path.join(test_hr_image_dir, x) for x in names]

    def __getitem__(self, batch_index: int) -> dict:
        lr = imgproc.read_image_rgb(self.lr_image_file_names[batch_index])
        hr = imgproc.read_image_rgb(self.hr_image_file_names[batch_index])
        return {"lr": imgproc.image_to_tensor(lr, False, False), "hr": imgproc.image_to_tensor(hr, False, False)}

    def __len__(self) -> int:
        return len(self.lr_image_file_names)


class CUDAPrefetcher:
    """The reference's batch source for the train / validation loops (`next()` -> batch dict or None, `reset()`, `len()`;
    reference dataset.py:271-312) as a small HBM staging pipeline: while the step of batch i runs, batch i+1 is uploaded
    on a side HIP stream from pinned memory; `next()` orders the consumer behind that upload with an event (no stream-wide
    wait) and tells the caching allocator which stream now uses the tensors.  The device work of a batch (degradation)
    starts one slot later, in `degrade.DegradationPrefetcher` / the train step."""

    def __init__(self, dataloader: DataLoader, device: torch.device) -> None:
        self.original_dataloader = dataloader
        self.device = torch.device(device)
        self._side = torch.cuda.Stream(device=self.device)
        self._it = None
        self._staged = None          # (batch dict on the device, upload-done event) or None at the end of the epoch
        self.reset()

    def _stage(self) -> None:
        try:
            host = next(self._it)
        except StopIteration:
            self._staged = None
            return
        with torch.cuda.stream(self._side):
            dev = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in host.items()}
            done = torch.cuda.Event()
            done.record(self._side)
        self._staged = (dev, done)

    def next(self):
        if self._staged is None:
            return None
        batch, done = self._staged
        consumer = torch.cuda.current_stream(self.device)
        consumer.wait_event(done)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(consumer)
        self._stage()
        return batch

    def reset(self) -> None:
        self._it = iter(self.original_dataloader)
        self._stage()

    def __len__(self) -> int:
        return len(self.original_dataloader)


class PrefetchGenerator:
    """Reference dataset.py:200-228: iterate `generator` on a daemon thread, `num_data_prefetch_queue` items ahead."""

    def __init__(self, generator, num_data_prefetch_queue: int) -> None:
        import queue
        import threading
        self._queue = queue.Queue(num_data_prefetch_queue)
        self._end = object()

        def work():
            for item in generator:
                self._queue.put(item)
            self._queue.put(self._end)
        self._thread = threading.Thread(target=work, daemon=True)
        self._thread.start()

    def __next__(self):
        item = self._queue.get()
        if item is self._end:
            raise StopIteration
        return item

    def __iter__(self):
        return self


class PrefetchDataLoader(DataLoader):
    """Reference dataset.py:231-245: a DataLoader whose iterator runs `num_data_prefetch_queue` batches ahead on a thread."""

    def __init__(self, num_data_prefetch_queue: int, **kwargs) -> None:
        self.num_data_prefetch_queue = num_data_prefetch_queue
        super().__init__(**kwargs)

    def __iter__(self):
        return PrefetchGenerator(super().__iter__(), self.num_data_prefetch_queue)


class CPUPrefetcher:
    """Reference dataset.py:248-268: `next()` -> batch or None, `reset()`, `len()` over a DataLoader, host side only."""

    def __init__(self, dataloader: DataLoader) -> None:
        self.original_dataloader = dataloader
        self.data = iter(dataloader)

    def next(self):
        try:
            return next(self.data)
        except StopIteration:
            return None

    def reset(self) -> None:
        self.data = iter(self.original_dataloader)

    def __len__(self) -> int:
        return len(self.original_dataloader)
