#!/usr/bin/env python
"""Stand-alone kernel times under a COLD protocol (task 2): every launch works on a different buffer set out of a ring
whose footprint exceeds the 256-MB Infinity Cache several times over, and the GEMMs carry the model's epilogues (fc1: GELU with both
stores, fc2 / proj: bias + residual, fc2-dX: times the saved gelu').  Device-side durations: run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -o g -- python3 tools/cold_bench.py [rows] [reps]
and summarise with tools/ktrace.py, or read the HIP-event times this script prints (launch gaps included: upper bounds).
rows: 12608 (full batch), 4334 (a third: one of three image chains), 2048 (text tower)."""
import os as _os; _os.environ.setdefault("FC_PROBES_LIB", "1")      # the fused-MLP entry points live in the tools build
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12608
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
D, Hd, H = 384, 1536, 6
N_tok = 197 if M % 197 == 0 else 32
B = M // N_tok
dev = "cuda"
bf = torch.bfloat16
GB = float(os.environ.get("COLD_GB", "1.5"))

def ring(shape, dtype=bf, scale=1.0):
    n = 1
    for s in shape: n *= s
    per = n * (2 if dtype == bf else 4)
    k = max(2, int(GB * 1e9 / 8 / per) + 1)      # each tensor family gets an eighth of the footprint budget at least
    return [(torch.randn(*shape, device=dev) * scale).to(dtype) for _ in range(min(k, 64))]

sp = _lib.stream_ptr()
g = torch.Generator(device="cpu").manual_seed(0)
X = ring((M, D)); X2 = ring((M, D)); Y = ring((M, D)); Y2 = ring((M, D))
QKV = ring((M, 3 * D)); DQKV = ring((M, 3 * D))
U = ring((M, Hd)); U2 = ring((M, Hd)); U3 = ring((M, Hd))
Wqkv = ring((3 * D, D), scale=D ** -0.5); Wp = ring((D, D), scale=D ** -0.5); W1 = ring((Hd, D), scale=D ** -0.5); W2 = ring((D, Hd), scale=Hd ** -0.5)
b384 = torch.randn(D, device=dev) * 0.1; b1152 = torch.randn(3 * D, device=dev) * 0.1; b1536 = torch.randn(Hd, device=dev) * 0.1
gam = torch.rand(D, device=dev) + 0.5; bet = torch.randn(D, device=dev) * 0.1
mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
lse = torch.zeros(B * H * N_tok, device=dev); delta = torch.zeros(B * H * N_tok, device=dev)
part = torch.empty(int(L.fc_k_layernorm_partial_floats(M, D)) + 64, device=dev)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
ck = _lib.check
def R(lst, i): return P(lst[i % len(lst)])
K = {}
K["ln_fwd"] = lambda i: ck(L.fc_k_layernorm_fwd(1, R(X, i), P(gam), P(bet), R(Y, i), P(mean), P(rstd), M, D, 1e-5, sp))
K["qkv fwd (bias)"] = lambda i: ck(L.fc_k_gemm_epi(0, R(X, i), R(Wqkv, i), R(QKV, i), M, 3 * D, D, P(b1152), None, None, None, sp))
K["attn fwd"] = lambda i: ck(L.fc_k_attention_fwd(1, 1, R(QKV, i), R(Y, i), P(lse), B, N_tok, H, 64, 0.125, sp))
K["proj fwd (bias+res)"] = lambda i: ck(L.fc_k_gemm_epi(0, R(X, i), R(Wp, i), R(Y, i), M, D, D, P(b384), R(X2, i), None, None, sp))
K["fc1 fwd (gelu, 2 stores)"] = lambda i: ck(L.fc_k_gemm_epi(0, R(X, i), R(W1, i), R(U, i), M, Hd, D, P(b1536), None, R(U2, i), None, sp))
K["fc2 fwd (bias+res)"] = lambda i: ck(L.fc_k_gemm_epi(0, R(U, i), R(W2, i), R(Y, i), M, D, Hd, P(b384), R(X2, i), None, None, sp))
K["fc2 dX (x gelu')"] = lambda i: ck(L.fc_k_gemm_epi(1, R(X, i), R(W2, i), R(U, i), M, Hd, D, None, None, None, R(U2, i), sp))
K["fc1 dX (plain)"] = lambda i: ck(L.fc_k_gemm_epi(1, R(U, i), R(W1, i), R(Y, i), M, D, Hd, None, None, None, None, sp))
K["ln_bwd"] = lambda i: ck(L.fc_k_layernorm_bwd_partial(1, R(Y, i), R(X, i), P(mean), P(rstd), P(gam), R(X2, i), R(Y2, i), P(dg), P(db), M, D, P(part), sp))
K["proj dX (plain)"] = lambda i: ck(L.fc_k_gemm_epi(1, R(X, i), R(Wp, i), R(Y, i), M, D, D, None, None, None, None, sp))
K["attn bwd"] = lambda i: ck(L.fc_k_attention_bwd(1, 1, R(QKV, i), R(X, i), R(Y, i), P(lse), P(delta), R(DQKV, i), B, N_tok, H, 64, 0.125, sp))
K["qkv dX (plain)"] = lambda i: ck(L.fc_k_gemm_epi(1, R(QKV, i), R(Wqkv, i), R(Y, i), M, D, 3 * D, None, None, None, None, sp))
DU = ring((M, Hd))
PF = [torch.empty(2 * D * Hd, device=dev, dtype=bf) for _ in W1]; PB = [torch.empty(2 * D * Hd, device=dev, dtype=bf) for _ in W1]
for a_, b_, c_, d_ in zip(W1, W2, PF, PB): ck(L.fc_k_mlp_pack(P(a_), P(b_), P(c_), P(d_), D, Hd, sp))
K["mlp fused fwd (fc1+gelu+fc2+res)"] = lambda i: ck(L.fc_k_mlp_fused(0, R(X, i), R(PF, i), P(b1536), P(b384), R(U, i), R(U2, i), R(X2, i), None, N_tok, R(Y, i), M, D, Hd, sp))
K["mlp fused bwd (fc2dX*g'+fc1dX)"] = lambda i: ck(L.fc_k_mlp_fused(1, R(X, i), R(PB, i), None, None, R(DU, i), R(U2, i), None, None, 1, R(Y, i), M, D, Hd, sp))
only = os.environ.get("COLD_ONLY")
flops = {"qkv fwd (bias)": 2 * M * 3 * D * D, "proj fwd (bias+res)": 2 * M * D * D, "fc1 fwd (gelu, 2 stores)": 2 * M * Hd * D,
         "fc2 fwd (bias+res)": 2 * M * Hd * D, "fc2 dX (x gelu')": 2 * M * Hd * D, "fc1 dX (plain)": 2 * M * Hd * D,
         "proj dX (plain)": 2 * M * D * D, "qkv dX (plain)": 2 * M * 3 * D * D,
         "mlp fused fwd (fc1+gelu+fc2+res)": 4 * M * Hd * D, "mlp fused bwd (fc2dX*g'+fc1dX)": 4 * M * Hd * D,
         "attn fwd": 4 * B * H * N_tok * N_tok * 64, "attn bwd": 10 * B * H * N_tok * N_tok * 64}
print(f"# rows {M} ({B} x {N_tok}), {reps} launches per kernel, each on another buffer set (ring footprint ~{GB} GB per tensor group of 8); HIP-event time per launch")
for name, fn in K.items():
    if only and only not in name: continue
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(3 + i)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tf = f"{flops[name] / us / 1e6:7.1f} TF/s" if name in flops else ""
    print(f"{name:28s} {us:8.1f} us  {tf}")
