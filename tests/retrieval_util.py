"""Synthetic retrieval sets (Flickr30k-shaped: 5 captions per image) for the N1 tests and golden vectors."""
from __future__ import annotations

import math

import torch

from synth import det_tensor


def unit(t: torch.Tensor) -> torch.Tensor:
    return t / t.norm(dim=-1, keepdim=True)


def retrieval_set(n_images: int, caps_per_image: int, D: int, seed: int, noise: float = 0.8):
    """image features [n_images, D], caption features [n_images*caps, D] (float32, unit norm), image ids, annotation ids."""
    img = unit(det_tensor([n_images, D], seed, 1.0))
    nz = det_tensor([n_images * caps_per_image, D], seed + 17, 1.0)
    cap = unit(img.repeat_interleave(caps_per_image, 0) + noise * nz)
    image_ids = 1000 + 7 * torch.arange(n_images, dtype=torch.int64)
    ann_ids = 50000 + 3 * torch.arange(n_images * caps_per_image, dtype=torch.int64)
    return img, cap, image_ids, ann_ids


def stream(img, cap, image_ids, ann_ids, caps_per_image: int, batch: int, mult: int = 37, add: int = 11):
    """Loader-order batches (images, captions, image_ids, ann_ids, index): a fixed pseudo-random permutation of the
    caption indices (the reference evaluates with a shuffling DataLoader, fedavgserver.py:687)."""
    n = cap.shape[0]
    assert math.gcd(mult, n) == 1
    perm = (torch.arange(n, dtype=torch.int64) * mult + add) % n
    out = []
    for s in range(0, n, batch):
        j = perm[s:s + batch]
        i = j // caps_per_image
        out.append((img[i].clone(), cap[j].clone(), image_ids[i].clone(), ann_ids[j].clone(), j.clone()))
    return out


class FakeDataset:
    def __init__(self, n_images, n_captions, iid_to_cls=None):
        self.n_images = n_images
        self.n = n_captions
        self.iid_to_cls = iid_to_cls

    def __len__(self):
        return self.n


class FakeLoader:
    def __init__(self, batches, dataset):
        self.batches = batches
        self.dataset = dataset

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class PassThroughModel:
    """model([images, captions], feat_out=True) -> [images, captions]: the 'features' ARE the inputs."""

    def __init__(self, D):
        self.embed_dim = D

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, x, feat_out=False):
        return [x[0], x[1]]


RETRIEVAL_CASES = {
    "flickr_like": dict(n_images=40, caps=5, D=16, seed=5, batch=32, folds=2, ipf=20, cpf=100),
    "wide": dict(n_images=24, caps=5, D=48, seed=9, batch=16, folds=3, ipf=8, cpf=40),
}
