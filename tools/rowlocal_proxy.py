#!/usr/bin/env python
"""Round 6, VERDICT r05 item 2 (row-local fused stage per layer: proj + res -> LN2 -> fc1 -> GELU -> fc2 + res -> LN1(next) -> qkv(next) in ONE
launch per 64-row panel): the kill criterion ("> 75 us cold at chain size: stop") priced BEFORE building, with the fused-MLP kernel that
exists (fc_mlp.hip, k_mlp_fused v4e) as the proxy.

Per row the stage multiplies by 384 x (384 + 1536 + 1536 + 1152) = 384 x 4608 weights; an MLP with hidden width 2304 multiplies by
384 x 2 x 2304 = the same 384 x 4608: the same FLOPs, the same 3.54 MB of weights streamed per panel, the same panel structure (activations
through LDS, packed weights through the register ring).  The proxy under-counts the stage's stores (the stage keeps 5 760 columns per row
for the backward -- gelu, gelu', qkv, h1, h2, xmid, x -- the proxy 4 992), leaves out both LayerNorms, and over-counts the GELU arithmetic
(2 304 instead of 1 536 activations per row).  Printed beside it: the MLP as it is (hidden 1 536) and the six separate kernels the stage
would replace, all under the cold protocol of tools/cold_bench.py (every launch on another buffer set).
    python tools/rowlocal_proxy.py [rows]          (HIP events; run under rocprofv3 --kernel-trace for device-side durations)"""
import os as _os; _os.environ.setdefault("FC_PROBES_LIB", "1")      # the fused-MLP entry points live in the tools build
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr; ck = _lib.check
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4334
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
D = 384
N_tok = 197 if M % 197 == 0 else 32
dev, bf = "cuda", torch.bfloat16
sp = _lib.stream_ptr()


def ring(shape, k, scale=1.0):
    return [(torch.randn(*shape, device=dev) * scale).to(bf) for _ in range(k)]


def timed(name, fn, flop):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(3 + i)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:52s} {us:8.1f} us   {flop / us / 1e6:7.1f} TFLOP/s")
    return us


K = 28                                                   # buffer sets per tensor family: 28 x (4334 x 2304 x 2 B = 20 MB) = 560 MB > 2 x the Infinity Cache
X, X2, Y = ring((M, D), K), ring((M, D), K), ring((M, D), K)
print(f"# rows {M}, {reps} launches per kernel, each on another buffer set; HIP-event time per launch (launch gaps included)")
tot = {}
for Hd in (1536, 2304):
    U, U2 = ring((M, Hd), K), ring((M, Hd), K)
    W1, W2 = ring((Hd, D), K, D ** -0.5), ring((D, Hd), K, Hd ** -0.5)
    PF = [torch.empty(2 * D * Hd, device=dev, dtype=bf) for _ in W1]; PB = [torch.empty(2 * D * Hd, device=dev, dtype=bf) for _ in W1]
    for a_, b_, c_, d_ in zip(W1, W2, PF, PB):
        ck(L.fc_k_mlp_pack(P(a_), P(b_), P(c_), P(d_), D, Hd, sp))
    b1 = torch.randn(Hd, device=dev) * 0.1; b2 = torch.randn(D, device=dev) * 0.1
    R = lambda lst, i: P(lst[i % len(lst)])
    tot[("f", Hd)] = timed(f"fused fwd panel kernel, hidden {Hd}" + (" (= the stage's FLOPs and weight bytes)" if Hd == 2304 else " (the MLP as it is)"),
                           lambda i: ck(L.fc_k_mlp_fused(0, R(X, i), R(PF, i), P(b1), P(b2), R(U, i), R(U2, i), R(X2, i), None, N_tok, R(Y, i), M, D, Hd, sp)),
                           4 * M * Hd * D)
    tot[("b", Hd)] = timed(f"fused bwd panel kernel, hidden {Hd}",
                           lambda i: ck(L.fc_k_mlp_fused(1, R(X, i), R(PB, i), None, None, R(U, i), R(U2, i), None, None, 1, R(Y, i), M, D, Hd, sp)),
                           4 * M * Hd * D)
    if Hd == 1536:
        Wq, Wp = ring((3 * D, D), K, D ** -0.5), ring((D, D), K, D ** -0.5)
        Q = ring((M, 3 * D), K)
        bq = torch.randn(3 * D, device=dev) * 0.1
        gam = torch.rand(D, device=dev) + 0.5; bet = torch.randn(D, device=dev) * 0.1
        mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
        parts = [
            ("proj fwd (bias + res)", lambda i: ck(L.fc_k_gemm_epi(0, R(X, i), R(Wp, i), R(Y, i), M, D, D, P(b2), R(X2, i), None, None, sp)), 2 * M * D * D),
            ("LayerNorm fwd", lambda i: ck(L.fc_k_layernorm_fwd(1, R(X, i), P(gam), P(bet), R(Y, i), P(mean), P(rstd), M, D, 1e-5, sp)), 0),
            ("fc1 fwd (gelu, 2 stores)", lambda i: ck(L.fc_k_gemm_epi(0, R(X, i), R(W1, i), R(U, i), M, Hd, D, P(b1), None, R(U2, i), None, sp)), 2 * M * Hd * D),
            ("fc2 fwd (bias + res)", lambda i: ck(L.fc_k_gemm_epi(0, R(U, i), R(W2, i), R(Y, i), M, D, Hd, P(b2), R(X2, i), None, None, sp)), 2 * M * Hd * D),
            ("qkv fwd (bias)", lambda i: ck(L.fc_k_gemm_epi(0, R(X, i), R(Wq, i), R(Q, i), M, 3 * D, D, P(bq), None, None, None, sp)), 2 * M * 3 * D * D),
        ]
        s = 0.0
        for name, fn, fl in parts:
            us = timed("  separate: " + name, fn, max(fl, 1))
            s += us * (2 if name.startswith("LayerNorm") else 1)
        tot["sep"] = s
        del Wq, Wp, Q
    del U, U2, W1, W2, PF, PB
    torch.cuda.empty_cache()
print(f"# six separate kernels (LayerNorm twice): {tot['sep']:.1f} us of kernel time + 6 dependent-launch boundaries (~5 us each in the step)")
print(f"# proxy of the fused forward stage: {tot[('f', 2304)]:.1f} us  (kill criterion: > 75 us cold)")
print(f"# proxy of the fused backward stage (fc2-dX x gelu' -> fc1-dX -> LN2-bwd -> proj-dX and qkv-dX -> LN1-bwd: the same 384 x 4608 per row): {tot[('b', 2304)]:.1f} us")
