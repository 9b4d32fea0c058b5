#!/usr/bin/env python
"""Micro-benchmark of the MFMA GEMM kinds on the client step's shapes (development aid)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
shapes = [("NT fc1", 0, 12608, 1536, 384), ("NT qkv", 0, 12608, 1152, 384), ("NT proj", 0, 12608, 384, 384), ("NT fc2", 0, 12608, 384, 1536),
          ("NN dfc2", 1, 12608, 1536, 384), ("NN dfc1", 1, 12608, 384, 1536), ("NN dqkv", 1, 12608, 384, 1152), ("TN dWfc1", 2, 1536, 384, 12608),
          ("NT txt fc1", 0, 2048, 1536, 384), ("NT txt proj", 0, 2048, 384, 384)]
if os.environ.get("GEMM_SHAPES"):   # "kind,M,N,K;kind,M,N,K"
    shapes = [("custom", *map(int, t.split(","))) for t in os.environ["GEMM_SHAPES"].split(";")]
    only = None
custom = bool(os.environ.get("GEMM_SHAPES"))   # custom shapes run without a bias: the PLAIN epilogue, as bench.py's roofline kernel
sp = _lib.stream_ptr()
for name, kind, M, N, K in shapes:
    if only and only not in name: continue
    shpA = (M, K) if kind != 2 else (K, M)
    shpB = (N, K) if kind == 0 else (K, N)
    # cold protocol (default; GEMM_WARM=1: one buffer set as in rounds 1-3): every launch reads another A / B and writes another C out of
    # a >= 1-GB ring, so nothing is served from the 256-MB Infinity Cache or an L2 that a previous launch of the loop filled
    out_dt = torch.float32 if kind == 2 else torch.bfloat16
    nA = 1
    for d_ in shpA: nA *= d_
    nset = 1 if os.environ.get("GEMM_WARM") else max(2, int(1e9 / (2 * nA)) + 1)
    As = [torch.randn(*shpA, device="cuda").bfloat16() for _ in range(nset)]
    Bs = [torch.randn(*shpB, device="cuda").bfloat16() for _ in range(min(nset, 12))]
    Cs = [torch.empty(M, N, device="cuda", dtype=out_dt) for _ in range(nset)]
    bias = torch.randn(N, device="cuda")
    def go(i):
        _lib.check(L.fc_k_gemm(1, kind, 1, 0 if kind == 2 else 1, P(As[i % nset]), P(Bs[i % len(Bs)]), P(Cs[i % nset]), M, N, K, None if custom else P(bias), 0, sp))
    for i in range(3):
        go(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        go(3 + i)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:12s} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s  ({'warm' if nset == 1 else 'cold: ring of %d sets' % nset})")
    del As, Bs, Cs
