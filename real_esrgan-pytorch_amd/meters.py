"""Console meters with the reference's names and call pattern (`Summary`, `AverageMeter(name, fmt, summary_type)`,
`ProgressMeter(num_batches, meters, prefix).display(batch)`; reference train_realesrnet.py:497-564) -- every batch counts in
the running average, as there -- without the reference's per-batch `.item()` synchronisation: `update` also takes a
0-d DEVICE tensor, which is accumulated on the device; the host only reads when a value is formatted (`display`, `.avg`,
`.val`), i.e. once per `print_frequency` batches.
"""
from __future__ import annotations

from enum import Enum
from typing import Iterable, Union

import torch

Number = Union[float, int, torch.Tensor]


class Summary(Enum):
    NONE = 0
    AVERAGE = 1
    SUM = 2
    COUNT = 3


def _host(v: Number) -> float:
    return float(v.item()) if torch.is_tensor(v) else float(v)


class AverageMeter:
    def __init__(self, name: str, fmt: str = ":f", summary_type: Summary = Summary.AVERAGE) -> None:
        self.name, self.fmt, self.summary_type = name, fmt, summary_type
        self.reset()

    def reset(self) -> None:
        self._val: Number = 0.0
        self._sum: Number = 0.0
        self.count = 0

    def update(self, val: Number, n: int = 1) -> None:
        if torch.is_tensor(val):
            val = val.detach().float()
            self._sum = self._sum + val * n if torch.is_tensor(self._sum) else val * n + self._sum   # stays on the device: no sync
        else:
            self._sum = self._sum + val * n
        self._val = val
        self.count += n

    # formatted values: the only places that read the device
    @property
    def val(self) -> float:
        return _host(self._val)

    @property
    def sum(self) -> float:
        return _host(self._sum)

    @property
    def avg(self) -> float:
        return self.sum / max(1, self.count)

    def __str__(self) -> str:
        spec = self.fmt.lstrip(":")
        return f"{self.name} {format(self.val, spec)} ({format(self.avg, spec)})"

    def summary(self) -> str:
        if self.summary_type is Summary.NONE:
            return ""
        value = {Summary.AVERAGE: self.avg, Summary.SUM: self.sum, Summary.COUNT: float(self.count)}[self.summary_type]
        return f"{self.name} {value:.2f}"


class ProgressMeter:
    def __init__(self, num_batches: int, meters: Iterable[AverageMeter], prefix: str = "") -> None:
        width = len(str(int(num_batches)))
        self._batch = lambda b: f"[{b:{width}d}/{num_batches}]"
        self.meters, self.prefix = list(meters), prefix

    def display(self, batch: int) -> None:
        print("\t".join([self.prefix + self._batch(batch)] + [str(m) for m in self.meters]))

    def display_summary(self) -> None:
        print(" ".join([" *"] + [m.summary() for m in self.meters]))
