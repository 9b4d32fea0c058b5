"""Helpers shared by the golden-vector tests."""
import json
import os

import torch

from synth import check_summary, det_ids, det_state_dict, det_tensor

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def init_record(case):
    for r in load("init.json"):
        if r["name"] == case:
            return r
    raise KeyError(case)


def case_shapes(case):
    return {k: tuple(s) for k, s in init_record(case)["keys"]}


def case_weights(case, base_seed=0):
    return det_state_dict(case_shapes(case), base_seed)


def case_inputs(rec):
    B, seq = rec["B"], rec["seq"]
    img = det_tensor((B, 3, 224, 224), 1000, 0.5)
    ids = det_ids((B, seq), 7, rec["mk"]["vocab_size"])
    nc = [x for x in rec["mk"]["num_classes"] if x]
    y = (torch.arange(B) * 3 + 1) % (max(nc) if nc else 1)
    return img, ids, y


def compare(t, ref, rtol, atol, what):
    if ref is None:
        assert t is None, what
        return
    if "full" in ref:
        exp = torch.tensor(ref["full"], dtype=torch.float64).reshape(ref["shape"])
        got = t.detach().double().cpu().reshape(ref["shape"])
        scale = float(exp.abs().max()) if exp.numel() else 0.0
        err = float((got - exp).abs().max()) if exp.numel() else 0.0
        assert err <= atol + rtol * scale, f"{what}: max err {err} (scale {scale})"
    else:
        check_summary(t.detach().cpu(), ref, rtol, atol, what)


def _vals(ref):
    """(expected values, positions or None) of a packed record."""
    if "full" in ref:
        return torch.tensor(ref["full"], dtype=torch.float64), None
    n = ref["numel"]
    from synth import SAMPLE_N
    pos = (torch.arange(SAMPLE_N, dtype=torch.int64) * 2654435761 + 97) % n
    return torch.tensor(ref["samples"], dtype=torch.float64), pos


def compare_after_adamw(t, ref_after, ref_grad, lr, what, g_rtol=2e-4, g_atol=1e-7, eps=1e-8):
    """Post-AdamW weights.  The first Adam step is lr*g/(|g|+eps): where |g| ~ eps (e.g. the k-bias, whose true
    gradient is 0) the update is round-off-sign dependent, so the tolerance is derived per element from the
    gradient tolerance: |d upd| <= lr*eps*dg/(|g|+eps)^2, capped at 2*lr."""
    exp, pos = _vals(ref_after)
    got = t.detach().double().cpu().reshape(-1)
    if pos is not None:
        got = got[pos]
    if ref_grad is None:
        tol = torch.full_like(exp, 1e-7)
    else:
        g, _ = _vals(ref_grad)
        rms = float(g.pow(2).mean().sqrt())
        dg = g_atol + g_rtol * torch.maximum(g.abs(), torch.tensor(rms, dtype=torch.float64))
        tol = 1e-7 + 2e-6 * exp.abs() + lr * torch.clamp(eps * dg / (g.abs() + eps) ** 2, max=2.0)
    bad = (got - exp).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off; worst {float(((got - exp).abs() - tol).max())}"
