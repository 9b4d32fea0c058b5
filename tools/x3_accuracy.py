#!/usr/bin/env python
"""Accuracy of the fp32 mode's split-operand MFMA GEMM (fc_gemm_x3.hip) against an fp64 product, beside the VALU kernel (fc_generic.hip) and
torch's fp32 matmul, on the model's shapes: max |err| / max |ref| and the RMS error relative to the RMS of the result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
sp = _lib.stream_ptr()
g = torch.Generator().manual_seed(0)
def run(kind, M, N, K, scaleA=0.5, scaleB=0.5, mean=0.0):
    shpA = (M, K) if kind != 2 else (K, M)
    shpB = (N, K) if kind == 0 else (K, N)
    A = (torch.randn(*shpA, generator=g) * scaleA + mean).cuda(); B = (torch.randn(*shpB, generator=g) * scaleB + mean).cuda()
    a = A.double() if kind != 2 else A.double().t()
    b = B.double().t() if kind == 0 else B.double()
    ref = a @ b
    out = {}
    for name, impl in (("x3", 1), ("valu", 0)):
        C = torch.zeros(M, N, device="cuda")
        rc = L.fc_k_gemm(impl, kind, 0, 0, P(A), P(B), P(C), M, N, K, None, 0, sp)
        assert rc == 0, rc
        torch.cuda.synchronize()
        out[name] = C.double()
    out["torch"] = ((A if kind != 2 else A.t()) @ (B.t() if kind == 0 else B)).double()
    s = f"kind {kind} {M:6d} x {N:5d} x {K:6d} mean {mean}:"
    for name, C in out.items():
        d = C - ref
        s += f"  {name} max {float(d.abs().max() / ref.abs().max()):.2e} rms {float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} bias {float(d.mean() / ref.abs().mean()):+.1e}"
    print(s)
run(0, 4334, 384, 384); run(0, 4334, 384, 1536); run(1, 4334, 384, 1536); run(2, 384, 1536, 12608); run(2, 1152, 384, 12608)
run(0, 4334, 384, 1536, mean=0.3); run(2, 384, 1536, 12608, mean=0.3)
