"""Seeding and metric bookkeeping with the reference's semantics (/root/reference/src/utils.py:35-46, 320-362)."""
from __future__ import annotations

import os
import random
from collections import defaultdict

import numpy as np
import torch


def set_seed(seed: int):
    """src/utils.py:35-46 (same call order: torch, random, numpy)."""
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


class Acc1:
    """Top-1 accuracy (src/metrics/metricszoo.py Acc1): collected on device, summarised once."""

    def __init__(self):
        self.correct, self.total = None, 0

    def collect(self, pred, true):
        c = (pred.argmax(dim=-1) == true).sum()
        self.correct = c if self.correct is None else self.correct + c
        self.total += int(true.numel())

    def summarize(self):
        v = float(self.correct) / max(1, self.total) if self.correct is not None else 0.0
        self.correct, self.total = None, 0
        return v


_METRICS = {"acc1": Acc1}


class MetricManager:
    """src/utils.py:320-362.  ``track(loss, pred, true)`` accumulates loss*len(pred); ``aggregate(total_len, step)``
    divides by the dataset length.  ``loss`` may be a device scalar: it is only read in ``aggregate`` (no per-step sync)."""

    def __init__(self, eval_metrics):
        self.metric_funcs = {name: _METRICS[name]() for name in eval_metrics if name in _METRICS}
        self.figures = defaultdict(int)
        self._results = dict()

    def track(self, loss, pred=None, true=None):
        self.figures["loss"] = self.figures["loss"] + loss * len(pred)
        for module in self.metric_funcs.values():
            module.collect(pred, true)

    def add_loss_sum(self, loss_times_n):
        """Device-side accumulation: add an already len-weighted loss sum (a tensor; read lazily)."""
        self.figures["loss"] = self.figures["loss"] + loss_times_n

    def aggregate(self, total_len, curr_step=None):
        running = {name: module.summarize() for name, module in self.metric_funcs.items()}
        running["loss"] = float(self.figures["loss"]) / total_len
        res = {"loss": running["loss"], "metrics": {name: running[name] for name in self.metric_funcs.keys()}}
        if curr_step is not None:
            self._results[curr_step] = res
        else:
            self._results = res
        self.figures = defaultdict(int)

    @property
    def results(self):
        return self._results
