#!/usr/bin/env python
"""Round 6: the fp32 attention backward on synthetic inputs in several regimes (plain, value rows with a common component v0, larger logits): per-element
error, error of the column sums of dqkv, and |column sums of dK| / column sums of |dK| (zero in exact arithmetic).  Run once with FC_LIB_PATH set to a
build from before the delta change and once without, to compare the two forms (profiles/r06/parity_margins.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr; S = _lib.stream_ptr
def run(qkv, dout, B, N, H, d):
    D = H * d
    qd = qkv.cuda(); o = torch.zeros(B, N, D, device="cuda"); lse = torch.zeros(B, H, N, device="cuda")
    _lib.check(L.fc_k_attention_fwd(1, 0, P(qd), P(o), P(lse), B, N, H, d, d ** -0.5, S()))
    dq = torch.zeros(B, N, 3 * D, device="cuda"); delta = torch.zeros(B, H, N, device="cuda")
    _lib.check(L.fc_k_attention_bwd(1, 0, P(qd), P(o), P(dout.cuda()), P(lse), P(delta), P(dq), B, N, H, d, d ** -0.5, S()))
    torch.cuda.synchronize()
    return dq.cpu().double()
def ref(qkv, dout, B, N, H, d):
    q5 = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4).double()
    q, k, v = q5[0] * d ** -0.5, q5[1], q5[2]
    Pm = torch.softmax(q @ k.transpose(-2, -1), -1)
    dO = dout.double().reshape(B, N, H, d).transpose(1, 2)
    dP = dO @ v.transpose(-2, -1)
    dS = Pm * (dP - (dP * Pm).sum(-1, keepdim=True))
    dq = (dS @ k) * d ** -0.5; dk = dS.transpose(-2, -1) @ q; dv = Pm.transpose(-2, -1) @ dO
    return torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * d)
B, N, H, d = 2, 40, 12, 64; D = H * d
g = torch.Generator().manual_seed(1)
for name, qs, v0s, dos in (("plain", 1.0, 0.0, 0.5), ("v0=40", 1.0, 40.0, 0.5), ("qk x3", 3.0, 0.0, 0.5), ("qk x3, v0=10", 3.0, 10.0, 0.5), ("qk x6", 6.0, 0.0, 0.5), ("v0=5", 1.0, 5.0, 0.5)):
    qkv = torch.randn(B, N, 3 * D, generator=g).reshape(B, N, 3, H, d)
    qkv[:, :, 0] *= qs; qkv[:, :, 1] *= qs
    qkv[:, :, 2] += torch.randn(1, 1, H, d, generator=g) * v0s
    qkv = qkv.reshape(B, N, 3 * D).contiguous(); dout = torch.randn(B, N, D, generator=g) * dos
    got = run(qkv, dout, B, N, H, d); r = ref(qkv, dout, B, N, H, d)
    cs_err = ((got - r).reshape(-1, 3 * D).sum(0).abs().max() / r.reshape(-1, 3 * D).sum(0).abs().max()).item()
    el = ((got - r).abs().max() / r.abs().max()).item()
    dk = got[..., D:2 * D]
    ratio = float((dk.sum(1).abs().max(-1).values / dk.abs().sum(1).max(-1).values).max())
    print(f"{name:14s} per-element {el:.2e}  column sums of dqkv err {cs_err:.2e}  colsum(dK)/colsum|dK| {ratio:.2e}")
