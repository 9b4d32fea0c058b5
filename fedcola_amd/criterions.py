"""Criteria of the client step, looked up by name like the reference does in ``torch.nn.__dict__``
(/root/reference/src/criterions/__init__.py:3-8, src/server/fedavgserver.py:76-80, src/client/fedavgclient.py:23).

``ContrastiveLoss`` restates torchmultimodal's ``ContrastiveLossWithTemperature`` (un-vendored third party; single
process: no gather): logit_scale = log(1/0.07) clamped to [0, log 100]; a fresh module is built every step at
fedavgclient.py:95, so the temperature never trains.  Both modules run the fused HIP kernels and support autograd
through a thin Function (the fully fused path is ``fc_client_step``)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr


def contrastive_tau() -> float:
    ls = torch.tensor(math.log(1 / 0.07), dtype=torch.float32).clamp(math.log(1.0), math.log(100.0))
    return float(torch.exp(ls))


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, tau):
        a = a.contiguous().float()
        b = b.contiguous().float()
        B, D = a.shape
        L = _lib.lib()
        scratch = torch.empty(int(L.fc_contrastive_scratch_floats(B)), device=a.device)
        lossbuf = torch.zeros(2, device=a.device)
        da, db = torch.empty_like(a), torch.empty_like(b)
        check(L.fc_contrastive_loss_fwd_bwd(ptr(a), ptr(b), B, D, tau, ptr(scratch), scratch.numel(), ptr(lossbuf), ptr(da), ptr(db),
                                            _lib.stream_ptr()))
        ctx.save_for_backward(da, db)
        return lossbuf[1].clone()

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return g * da, g * db, None


class ContrastiveLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.tensor(math.log(1 / 0.07)))

    def forward(self, embeddings_a, embeddings_b):
        self.logit_scale.data.clamp_(math.log(1.0), math.log(100.0))
        return _ContrastiveFn.apply(embeddings_a, embeddings_b, float(torch.exp(self.logit_scale)))


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, y):
        logits = logits.contiguous().float()
        B, Cn = logits.shape
        lossbuf = torch.zeros(2, device=logits.device)
        dl = torch.empty_like(logits)
        check(_lib.lib().fc_ce_loss_fwd_bwd(ptr(logits), ptr(y.contiguous().long()), B, Cn, ptr(lossbuf), ptr(dl), _lib.stream_ptr()))
        ctx.save_for_backward(dl)
        return lossbuf[1].clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return g * dl, None


class CrossEntropyLoss(nn.Module):
    def forward(self, logits, targets):
        return _CEFn.apply(logits, targets)


CRITERIA = {"ContrastiveLoss": ContrastiveLoss, "CrossEntropyLoss": CrossEntropyLoss}
