#!/usr/bin/env python
"""Vendor yardstick (task 1).  TOOLS ONLY -- nothing here is imported by fedcola_amd/ or timed by bench.py's headline.

Leg 1: hipBLASLt / rocBLAS (behind torch.matmul) against this repo's hand-written MFMA GEMMs on the eight per-layer shapes of the ViT-S
client step (qkv, proj, fc1, fc2 forward; fc2-dX, fc1-dX, proj-dX, qkv-dX) x {12608, 4334, 2048} rows, plus the four TN weight-gradient
shapes, under the SAME cold ring as tools/cold_bench.py (every launch reads another A / B and writes another C out of a >= 1-GB ring).
Both sides run the plain product (no bias, no activation): `fc_k_gemm` with the PLAIN epilogue against `torch.matmul(out=...)`.
HIP-event time per launch is printed (launch gaps included: an upper bound); run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -o y -- python3 tools/vendor_yardstick.py gemm
and summarise with `KTRACE_BYNAME=1 python tools/ktrace.py DIR` for device-side durations and the vendor kernels' NAMES (they carry
the macro-tile, the wave layout and the unroll depth the vendor's heuristic picked).

Leg 2 (`step`): the same model written in plain PyTorch-ROCm (F.linear / F.layer_norm / F.scaled_dot_product_attention / F.gelu, autograd,
bf16 autocast over fp32 master weights, fused AdamW), whole client step at B = 64 -- the "what would eager PyTorch do on this box" number.

usage: tools/vendor_yardstick.py [gemm|step|all] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = "cuda"
bf = torch.bfloat16
D, Hd = 384, 1536


SEG = [0]


def ev_time(fn, reps, warm=3, label=None):
    # a marker kernel in front of every timed leg: `KTRACE_SPLIT=spin tools/ktrace.py DIR` then prints one segment per leg, in this order
    torch.cuda._sleep(2000); SEG[0] += 1
    if label: print(f"#   segment {SEG[0]}: {label}")
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(warm + i)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def gemm_leg():
    from fedcola_amd import _lib
    L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr()
    # (name, kind 0 NT / 1 NN / 2 TN, N, K) per row count M.  NT: C = A[M,K] W[N,K]^T; NN: C = A[M,K] W[K,N]; TN: C[N1,N2] = A[M,N1]^T B[M,N2]
    layer = [("qkv fwd", 0, 3 * D, D), ("proj fwd", 0, D, D), ("fc1 fwd", 0, Hd, D), ("fc2 fwd", 0, D, Hd),
             ("fc2 dX", 1, Hd, D), ("fc1 dX", 1, D, Hd), ("proj dX", 1, D, D), ("qkv dX", 1, D, 3 * D)]
    dw = [("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc1", Hd, D), ("dW fc2", D, Hd)]
    print(f"# cold ring (>= 1 GB of A per shape, 12 weight matrices), {reps} launches each, HIP-event us per launch (upper bound: launch gaps included)")
    print(f"# {'shape':10s} {'rows':>6s} {'N':>5s} {'K':>5s} | {'ours us':>8s} {'TF/s':>7s} | {'vendor us':>9s} {'TF/s':>7s} | vendor/ours")
    for M in (12608, 4334, 2048):
        for name, kind, N, K in layer:
            nset = max(2, int(1e9 / (2 * M * K)) + 1)
            As = [torch.randn(M, K, device=dev).to(bf) for _ in range(nset)]
            Ws = [(torch.randn((N, K) if kind == 0 else (K, N), device=dev) * K ** -0.5).to(bf) for _ in range(12)]
            Cs = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(nset)]
            ours = ev_time(lambda i: _lib.check(L.fc_k_gemm(1, kind, 1, 1, P(As[i % nset]), P(Ws[i % 12]), P(Cs[i % nset]), M, N, K, None, 0, sp)), reps, label=f'ours   {name} rows {M}')
            if kind == 0:
                vend = ev_time(lambda i: torch.matmul(As[i % nset], Ws[i % 12].t(), out=Cs[i % nset]), reps, label=f'vendor {name} rows {M}')
            else:
                vend = ev_time(lambda i: torch.matmul(As[i % nset], Ws[i % 12], out=Cs[i % nset]), reps, label=f'vendor {name} rows {M}')
            fl = 2.0 * M * N * K
            print(f"  {name:10s} {M:6d} {N:5d} {K:5d} | {ours:8.1f} {fl / ours / 1e6:7.1f} | {vend:9.1f} {fl / vend / 1e6:7.1f} | {vend / ours:5.2f}")
            del As, Ws, Cs
        for name, N1, N2 in dw:      # dW[N1,N2] = dY[M,N1]^T X[M,N2] (fp32 out on our side; the vendor side writes bf16 -- less store traffic, noted)
            nset = max(2, int(1e9 / (2 * M * (N1 + N2))) + 1)
            dY = [torch.randn(M, N1, device=dev).to(bf) for _ in range(nset)]
            X = [torch.randn(M, N2, device=dev).to(bf) for _ in range(nset)]
            Co = [torch.empty(N1, N2, device=dev) for _ in range(4)]
            Cv = [torch.empty(N1, N2, device=dev, dtype=bf) for _ in range(4)]
            db = torch.empty(N1, device=dev)
            ours = ev_time(lambda i: _lib.check(L.fc_k_dw(2, P(dY[i % nset]), P(X[i % nset]), P(Co[i % 4]), P(db), M, N1, N2, sp)), reps, label=f'ours   {name} rows {M}')
            vend = ev_time(lambda i: torch.matmul(dY[i % nset].t(), X[i % nset], out=Cv[i % 4]), reps, label=f'vendor {name} rows {M}')
            fl = 2.0 * M * N1 * N2
            print(f"  {name:10s} {M:6d} {N1:5d}x{N2:<5d}| {ours:8.1f} {fl / ours / 1e6:7.1f} | {vend:9.1f} {fl / vend / 1e6:7.1f} | {vend / ours:5.2f}   (ours: the entry point allocates and synchronises per call -- read OUR time from the kernel trace)")
            del dY, X, Co, Cv


class TorchTower(torch.nn.Module):
    def __init__(self, img, depth=12, heads=6, vocab=7732, seq=32):
        super().__init__()
        nn = torch.nn
        self.img, self.heads = img, heads
        if img:
            self.proj = nn.Conv2d(3, D, 16, 16)
            self.cls = nn.Parameter(torch.zeros(1, 1, D)); self.pos = nn.Parameter(torch.zeros(1, 197, D))
        else:
            self.word = nn.Embedding(vocab, D, padding_idx=0); self.tpos = nn.Embedding(seq, D); self.ttype = nn.Embedding(2, D)
            self.eln = nn.LayerNorm(D, eps=1e-12)
        self.n1 = nn.ModuleList(nn.LayerNorm(D) for _ in range(depth)); self.n2 = nn.ModuleList(nn.LayerNorm(D) for _ in range(depth))
        self.qkv = nn.ModuleList(nn.Linear(D, 3 * D) for _ in range(depth)); self.pr = nn.ModuleList(nn.Linear(D, D) for _ in range(depth))
        self.fc1 = nn.ModuleList(nn.Linear(D, Hd) for _ in range(depth)); self.fc2 = nn.ModuleList(nn.Linear(Hd, D) for _ in range(depth))

    def forward(self, x):
        if self.img:
            x = self.proj(x).flatten(2).transpose(1, 2)
            x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], 1) + self.pos
        else:
            n = x.shape[1]
            x = self.eln(self.word(x) + self.ttype.weight[0] + self.tpos.weight[:n])
        B, N, _ = x.shape
        for l in range(len(self.n1)):
            q, k, v = self.qkv[l](self.n1[l](x)).reshape(B, N, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
            o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)
            x = x + self.pr[l](o)
            x = x + self.fc2[l](F.gelu(self.fc1[l](self.n2[l](x))))
        return x[:, 0]


def step_leg():
    torch.manual_seed(0)
    B = 64
    ti, tt = TorchTower(True).to(dev), TorchTower(False).to(dev)
    norm = torch.nn.LayerNorm(D, eps=1e-6).to(dev)
    params = list(ti.parameters()) + list(tt.parameters()) + list(norm.parameters())
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.0, fused=True)
    img = (torch.randn(B, 3, 224, 224, device=dev) * 0.5).clamp_(-1, 1)
    ids = torch.randint(1, 7732, (B, 32), device=dev)
    tgt = torch.arange(B, device=dev)

    def one(_):
        with torch.autocast("cuda", dtype=bf):
            a = F.normalize(norm(ti(img)).float(), dim=-1); b = F.normalize(norm(tt(ids)).float(), dim=-1)
        lg = a @ b.t() * (1 / 0.07)
        loss = 0.5 * (F.cross_entropy(lg, tgt) + F.cross_entropy(lg.t(), tgt))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    us = ev_time(one, max(10, reps), warm=5)
    print(f"# plain PyTorch-ROCm (eager, bf16 autocast, SDPA, fused AdamW), ViT-S img+txt, B = 64: {us / 1e3:.2f} ms per step = {B / us * 1e6:.0f} pairs/s"
          f" ({2.023e12 / us / 1e6:.0f} TFLOP/s of the 2.023 algorithmic TFLOP)")


if what in ("gemm", "all"): gemm_leg()
if what in ("step", "all"): step_leg()
