"""Deterministic, library-independent synthetic tensors for tests and golden vectors.

Values come from exact int64 arithmetic (then one exact int->float conversion and one fp32 multiply),
so the generator reproduces bit-identically on any machine / torch build.
"""
from __future__ import annotations

import zlib
from typing import Dict, Sequence

import torch

_P = 65521  # prime < 2**16


def det_tensor(shape: Sequence[int], seed: int, scale: float = 1.0, offset: float = 0.0) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= int(s)
    idx = torch.arange(n, dtype=torch.int64)
    h = (idx * 48271 + (seed % 100003) * 69621 + (idx // 7) * 40692 + 12345) % _P
    h = (h * 1103 + (idx % 11) * 9973) % _P
    v = (h.to(torch.float32) - float(_P // 2)) * (2.0 / _P)          # in (-1, 1)
    return (v * scale + offset).reshape(tuple(shape))


def det_ids(shape: Sequence[int], seed: int, vocab: int, pad_tail: bool = True) -> torch.Tensor:
    """int64 token ids in [1, vocab-1]; BERT-like zero-padded tail of pseudo-random per-row length."""
    B, N = shape
    idx = torch.arange(B * N, dtype=torch.int64)
    ids = ((idx * 7919 + seed * 104729 + 17) % (vocab - 1)) + 1
    ids = ids.reshape(B, N)
    if pad_tail:
        for r in range(B):
            ln = max(2, N - ((r * 5 + seed) % max(1, N // 2)))
            ids[r, ln:] = 0
    return ids


def key_seed(key: str, base: int = 0) -> int:
    return (zlib.crc32(key.encode()) + base) % 100003


def det_state_dict(shapes: Dict[str, Sequence[int]], base_seed: int = 0) -> Dict[str, torch.Tensor]:
    """Reference-shaped weights with sensible magnitudes per parameter kind (all non-zero, incl. pos/cls)."""
    out = {}
    for k, shp in shapes.items():
        s = key_seed(k, base_seed)
        if k.endswith("position_ids"):
            out[k] = torch.arange(shp[-1], dtype=torch.int64).reshape(tuple(shp))
        elif "norm" in k.lower() and k.endswith("weight"):
            out[k] = det_tensor(shp, s, 0.2, 1.0)
        elif k.endswith("bias"):
            out[k] = det_tensor(shp, s, 0.05)
        elif "cross_modal_scale" in k:
            out[k] = det_tensor(shp, s, 0.3, 0.5)
        elif "word_embeddings" in k:
            w = det_tensor(shp, s, 0.5)
            w[0] = 0.0                                            # padding_idx row
            out[k] = w
        elif "embeddings" in k and "proj.weight" not in k:
            out[k] = det_tensor(shp, s, 0.3)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= int(d)
            out[k] = det_tensor(shp, s, 1.2 / max(1.0, fan_in) ** 0.5)
    return out


SAMPLE_N = 24


def summarize(t: torch.Tensor):
    """Compact fingerprint of a tensor: sum, L1, L2 and SAMPLE_N values at fixed pseudo-random positions."""
    f = t.detach().reshape(-1).to(torch.float64)
    n = f.numel()
    pos = (torch.arange(SAMPLE_N, dtype=torch.int64) * 2654435761 + 97) % n
    return dict(sum=float(f.sum()), l1=float(f.abs().sum()), l2=float(f.pow(2).sum().sqrt()),
                samples=[float(x) for x in f[pos]], numel=int(n))


def check_summary(t: torch.Tensor, ref: dict, rtol: float, atol: float, what: str = ""):
    s = summarize(t)
    assert s["numel"] == ref["numel"], f"{what}: numel {s['numel']} != {ref['numel']}"
    scale = ref["l2"] / max(1.0, ref["numel"]) ** 0.5            # rms magnitude
    for a, b in zip(s["samples"], ref["samples"]):
        assert abs(a - b) <= atol + rtol * max(abs(b), scale), f"{what}: sample {a} vs {b} (rms {scale})"
    assert abs(s["l2"] - ref["l2"]) <= atol + rtol * ref["l2"] * 4, f"{what}: l2 {s['l2']} vs {ref['l2']}"
    assert abs(s["l1"] - ref["l1"]) <= atol * ref["numel"] ** 0.5 + rtol * ref["l1"] * 4, f"{what}: l1 {s['l1']} vs {ref['l1']}"
