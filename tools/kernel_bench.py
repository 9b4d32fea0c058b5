#!/usr/bin/env python
"""Stand-alone launches of the non-GEMM kernels of the image chain at their in-model shapes (run under
`rocprofv3 --kernel-trace` + tools/ktrace.py for device-side durations; the printed event timings include ~4 us of launch each).
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
bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
qkv, o, do_, dqkv = bf(B, N, 3 * D), bf(B, N, D), bf(B, N, D), torch.empty(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda"); delta = torch.empty(B, H, N, device="cuda")
x, dy, res, y, dx = bf(M, D), bf(M, D), bf(M, D), torch.empty(M, D, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
g, b_ = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")


def run(name, f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us (events, incl. launch)")


run("attn fwd", lambda: _lib.check(L.fc_k_attention_fwd(1, 1, P(qkv), P(o), P(lse), B, N, H, d, d ** -0.5, sp)))
run("attn bwd", lambda: _lib.check(L.fc_k_attention_bwd(1, 1, P(qkv), P(o), P(do_), P(lse), P(delta), P(dqkv), B, N, H, d, d ** -0.5, sp)))
run("ln fwd", lambda: _lib.check(L.fc_k_layernorm_fwd(1, P(x), P(g), P(b_), P(y), P(mean), P(rstd), M, D, 1e-5, sp)))
run("ln bwd", lambda: _lib.check(L.fc_k_layernorm_bwd(1, P(dy), P(x), P(mean), P(rstd), P(g), P(res), P(dx), P(dg), P(db), M, D, sp)))
part = torch.empty(int(L.fc_k_layernorm_partial_floats(M, D)), device="cuda")
run("ln bwd partial", lambda: _lib.check(L.fc_k_layernorm_bwd_partial(1, P(dy), P(x), P(mean), P(rstd), P(g), P(res), P(dx), P(dg), P(db), M, D, P(part), sp)))
# the four weight-gradient problems of one layer through both grouped kernels (one problem per launch; in the step a chunk of 4
# layers x 2 towers is ONE launch, so these are per-problem device times, not the in-step cost)
for name, out, inn in (("dW qkv", 1152, 384), ("dW proj", 384, 384), ("dW fc1", 1536, 384), ("dW fc2", 384, 1536)):
    dYm, Xm = bf(M, out), bf(M, inn)
    dWm, dbm = torch.empty(out, inn, device="cuda"), torch.empty(out, device="cuda")
    for wide in (0, 1, 2):
        run(f"{name} wide={wide}", lambda: _lib.check(L.fc_k_dw(wide, P(dYm), P(Xm), P(dWm), P(dbm), M, out, inn, sp)))
