#!/usr/bin/env python
"""Stand-alone launches of the non-GEMM kernels of the image chain at their in-model shapes (run under
`rocprofv3 --kernel-trace` + tools/ktrace.py for device-side durations; the printed event timings include ~4 us of launch each).
Cold protocol (round 4; KERNEL_BENCH_WARM=1 restores the one-buffer-set loops of rounds 1-3): every launch works on another buffer set
out of a ring of >= 1 GB per kernel, so nothing is served from the 256-MB Infinity Cache or an L2 that the previous launch filled.
usage: tools/kernel_bench.py [reps] [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, H, d, D = 197, 6, 64, 384
M = B * N
sp = _lib.stream_ptr()
WARM = bool(os.environ.get("KERNEL_BENCH_WARM"))


def ring(*shape, empty=False):
    n = 2
    for d_ in shape: n *= d_
    k = 1 if WARM else max(2, min(48, int(0.25e9 / n) + 1))      # a quarter GB per tensor family: a kernel's four families pass 1 GB
    return [torch.empty(*shape, device="cuda", dtype=torch.bfloat16) if empty else torch.randn(*shape, device="cuda").bfloat16() for _ in range(k)]


R = lambda lst, i: P(lst[i % len(lst)])
qkv, o, do_, dqkv = ring(B, N, 3 * D), ring(B, N, D), ring(B, N, D), ring(B, N, 3 * D, empty=True)
lse = torch.empty(B, H, N, device="cuda"); delta = torch.empty(B, H, N, device="cuda")
x, dy, res, y, dx = ring(M, D), ring(M, D), ring(M, D), ring(M, D, empty=True), ring(M, D, empty=True)
g, b_ = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")


def run(name, f):
    for i in range(3): f(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): f(3 + i)
    e1.record(); e1.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us (events, incl. launch; {'warm' if WARM else 'cold ring'})")


run("attn fwd", lambda i: _lib.check(L.fc_k_attention_fwd(1, 1, R(qkv, i), R(o, i), P(lse), B, N, H, d, d ** -0.5, sp)))
run("attn bwd", lambda i: _lib.check(L.fc_k_attention_bwd(1, 1, R(qkv, i), R(o, i), R(do_, i), P(lse), P(delta), R(dqkv, i), B, N, H, d, d ** -0.5, sp)))
run("ln fwd", lambda i: _lib.check(L.fc_k_layernorm_fwd(1, R(x, i), P(g), P(b_), R(y, i), P(mean), P(rstd), M, D, 1e-5, sp)))
run("ln bwd", lambda i: _lib.check(L.fc_k_layernorm_bwd(1, R(dy, i), R(x, i), P(mean), P(rstd), P(g), R(res, i), R(dx, i), P(dg), P(db), M, D, sp)))
part = torch.empty(int(L.fc_k_layernorm_partial_floats(M, D)), device="cuda")
run("ln bwd partial", lambda i: _lib.check(L.fc_k_layernorm_bwd_partial(1, R(dy, i), R(x, i), P(mean), P(rstd), P(g), R(res, i), R(dx, i), P(dg), P(db), M, D, P(part), sp)))
del qkv, o, do_, dqkv, x, dy, res, y, dx
# the four weight-gradient problems of one layer through both grouped kernels (one problem per launch; in the step a chunk of 4
# layers x 2 towers is ONE launch, so these are per-problem device times, not the in-step cost)
for name, out, inn in (("dW qkv", 1152, 384), ("dW proj", 384, 384), ("dW fc1", 1536, 384), ("dW fc2", 384, 1536)):
    dYm, Xm = ring(M, out), ring(M, inn)
    dWm, dbm = torch.empty(out, inn, device="cuda"), torch.empty(out, device="cuda")
    for wide in (0, 1, 2):
        run(f"{name} wide={wide}", lambda i: _lib.check(L.fc_k_dw(wide, R(dYm, i), R(Xm, i), P(dWm), P(dbm), M, out, inn, sp)))
    del dYm, Xm
