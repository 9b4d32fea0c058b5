"""Shared synthetic fixtures of the CreamFL row (N2): toy models, public set, client datasets, global public features."""
from __future__ import annotations

import torch

from synth import det_ids, det_tensor

D, SEQ, VOCAB, P = 16, 8, 30, 10
MK = {
    "mm": dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=D, depth=1, num_heads=2,
               vocab_size=VOCAB, max_text_len=SEQ),
    "img": dict(modalities=["img", None], num_classes=[10, None], tasks=["cls", None], embed_dim=D, depth=1, num_heads=2,
                vocab_size=VOCAB, max_text_len=SEQ),
    "txt": dict(modalities=[None, "txt"], num_classes=[None, 4], tasks=[None, "cls"], embed_dim=D, depth=1, num_heads=2,
                vocab_size=VOCAB, max_text_len=SEQ),
}
BASE_SEED = {"mm": 11, "img": 12, "txt": 13}
CREAM_ARGS = dict(E=1, B=4, lr=1e-3, optimizer="AdamW", no_shuffle=True, interintra_weight=0.5, pub_batch_size=4, no_mm_contrastive=False,
                  kd_weight=0.3, p_lr=1e-3, algorithm="creamfl", compensation=False, vocab_size=VOCAB, seq_len=SEQ)


class PubSet(torch.utils.data.Dataset):
    """Public (image, caption) pairs; the last field is the dataset-level index the distillation dictionaries are keyed by."""

    def __init__(self, n=P):
        self.img = det_tensor((n, 3, 224, 224), 5000, 0.5)
        self.ids = det_ids((n, SEQ), 31, VOCAB)
        self.index = 100 + 3 * torch.arange(n)

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.ids[i], i, i, self.index[i]


class Pairs(torch.utils.data.Dataset):
    def __init__(self, n=6):
        self.img = det_tensor((n, 3, 224, 224), 2100, 0.5)
        self.ids = det_ids((n, SEQ), 17, VOCAB)

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.ids[i], i // 5, i, i


class Cls(torch.utils.data.Dataset):
    def __init__(self, kind, n=6, classes=10):
        self.x = det_tensor((n, 3, 224, 224), 3100, 0.5) if kind == "img" else det_ids((n, SEQ), 19, VOCAB)
        self.y = (torch.arange(n) * 7 + 1) % classes

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


def global_features():
    gi = det_tensor((P, D), 71, 1.0)
    gt = det_tensor((P, D), 72, 1.0)
    return gi / gi.norm(dim=-1, keepdim=True), gt / gt.norm(dim=-1, keepdim=True)


def client_pub_features(seed):
    f = det_tensor((P, D), seed, 1.0)
    return f / f.norm(dim=-1, keepdim=True)
