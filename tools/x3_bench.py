#!/usr/bin/env python
"""Stand-alone times of the fp32 mode's kernels on the model's shapes (HIP events, a ring of operand sets so that launches do not re-read a warm cache):
split-operand MFMA GEMMs (fc_gemm_x3.hip) and fp32 MFMA attention (fc_attn_f32.hip).  TFLOP/s are fp32-equivalent (2 M N K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
sp = _lib.stream_ptr()
def bench(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(3 + i)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def gemm(kind, M, N, K):
    shpA = (M, K) if kind != 2 else (K, M)
    shpB = (N, K) if kind == 0 else (K, N)
    ring = max(2, int(6e8 / (4 * (M * K + N * K + M * N))) + 1)
    A = [torch.randn(*shpA, device="cuda") for _ in range(ring)]; B = [torch.randn(*shpB, device="cuda") for _ in range(ring)]
    C = [torch.empty(M, N, device="cuda") for _ in range(ring)]
    us = bench(lambda i: _lib.check(L.fc_k_gemm(1, kind, 0, 0, P(A[i % ring]), P(B[i % ring]), P(C[i % ring]), M, N, K, None, 0, sp)))
    print(f"x3 gemm kind {kind} {M:6d} x {N:5d} x {K:6d}: {us:8.1f} us  {2 * M * N * K / us / 1e6:6.1f} TFLOP/s fp32-equivalent ({6 * 2 * M * N * K / us / 1e6:6.0f} TFLOP/s of bf16 MFMA)")
for kind, M, N, K in ((0, 12608, 1152, 384), (0, 12608, 384, 384), (0, 12608, 1536, 384), (0, 12608, 384, 1536), (1, 12608, 384, 1536), (1, 12608, 1536, 384),
                      (1, 12608, 384, 1152), (2, 384, 1536, 12608), (2, 1152, 384, 12608)):
    gemm(kind, M, N, K)
B, N, H = 64, 197, 6
qkv = [torch.randn(B, N, 3 * H * 64, device="cuda") for _ in range(8)]; o = torch.empty(B, N, H * 64, device="cuda"); lse = torch.empty(B, H, N, device="cuda")
do = torch.randn(B, N, H * 64, device="cuda"); dqkv = torch.empty(B, N, 3 * H * 64, device="cuda"); delta = torch.empty(B, H, N, device="cuda")
us = bench(lambda i: _lib.check(L.fc_k_attention_fwd(1, 0, P(qkv[i % 8]), P(o), P(lse), B, N, H, 64, 0.125, sp)))
print(f"fp32 attention fwd B {B} N {N} H {H}: {us:8.1f} us  {4 * B * H * N * N * 64 / us / 1e6:6.1f} TFLOP/s")
us = bench(lambda i: _lib.check(L.fc_k_attention_bwd(1, 0, P(qkv[i % 8]), P(o), P(do), P(lse), P(delta), P(dqkv), B, N, H, 64, 0.125, sp)))
print(f"fp32 attention bwd B {B} N {N} H {H}: {us:8.1f} us  {10 * B * H * N * N * 64 / us / 1e6:6.1f} TFLOP/s")
