#!/usr/bin/env python
"""Where does the fp32 mode's distance from the exact gradient come from?  ("or find the term".)
Each fp32-mode kernel at the 768-wide test's shapes against an fp64 evaluation of the same op, beside torch's own fp32 CPU op:
max |err| / max |ref| and rms err / rms ref.  A kernel whose figure is several times torch's is a term worth fixing."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr; ck = _lib.check
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(3)
_keep = []
def dev(t):
    _keep.append(t.float().cuda().contiguous())      # (kept alive: the pointer outlives the expression that made it)
    return _keep[-1]


def fig(name, ours, theirs, ref):
    ours, theirs, ref = ours.double().cpu(), theirs.double(), ref.double()
    mx = float(ref.abs().max()); rms = float(ref.pow(2).mean().sqrt())
    eo, et = ours - ref, theirs - ref
    print(f"{name:34s} library max {float(eo.abs().max()) / mx:.2e} rms {float(eo.pow(2).mean().sqrt()) / rms:.2e} | torch fp32 max {float(et.abs().max()) / mx:.2e} rms {float(et.pow(2).mean().sqrt()) / rms:.2e}"
          f" | ratio max {float(eo.abs().max()) / max(float(et.abs().max()), 1e-30):.1f} rms {float(eo.pow(2).mean().sqrt()) / max(float(et.pow(2).mean().sqrt()), 1e-30):.1f}")


D, Hd, H = 768, 3072, 12
for M, N_tok in ((320, 40), (1576, 197)):
    print(f"# rows {M} ({M // N_tok} x {N_tok}), D = {D}")
    x = torch.randn(M, D, generator=g); dy = torch.randn(M, D, generator=g) * 0.1; res = torch.zeros(M, D)
    gam = torch.randn(D, generator=g) * 0.2 + 1; bet = torch.randn(D, generator=g) * 0.1
    # LayerNorm
    y64 = torch.nn.functional.layer_norm(x.double(), (D,), gam.double(), bet.double(), 1e-5); y32 = torch.nn.functional.layer_norm(x, (D,), gam, bet, 1e-5)
    xd = dev(x); y = torch.empty_like(xd); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    ck(L.fc_k_layernorm_fwd(0, P(xd), P(dev(gam)), P(dev(bet)), P(y), P(mean), P(rstd), M, D, 1e-5, S()))
    fig("layernorm fwd", y, y32, y64)
    x64 = x.double().requires_grad_(True); torch.nn.functional.layer_norm(x64, (D,), gam.double(), bet.double(), 1e-5).backward(dy.double())
    x32 = x.clone().requires_grad_(True); torch.nn.functional.layer_norm(x32, (D,), gam, bet, 1e-5).backward(dy)
    dx = torch.empty_like(xd); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    ck(L.fc_k_layernorm_bwd(0, P(dev(dy)), P(xd), P(mean), P(rstd), P(dev(gam)), P(dev(res)), P(dx), P(dg), P(db), M, D, S()))
    fig("layernorm bwd dx", dx, x32.grad, x64.grad)
    # linears (NT fwd, NN dX, TN dW)
    for name, N, K in (("qkv", 3 * D, D), ("fc1", Hd, D), ("fc2", D, Hd)):
        A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * K ** -0.5; b = torch.randn(N, generator=g) * 0.1
        Cd = torch.empty(M, N, device="cuda")
        rc = L.fc_k_gemm(1, 0, 0, 0, P(dev(A)), P(dev(W)), P(Cd), M, N, K, P(dev(b)), 0, S()); assert rc == 0, rc
        fig(f"linear fwd {name} {N}x{K}", Cd, A @ W.t() + b, A.double() @ W.double().t() + b.double())
        G = torch.randn(M, N, generator=g) * 0.1
        Cd = torch.empty(M, K, device="cuda")
        rc = L.fc_k_gemm(1, 1, 0, 0, P(dev(G)), P(dev(W)), P(Cd), M, K, N, None, 0, S()); assert rc == 0, rc
        fig(f"linear dX  {name}", Cd, G @ W, G.double() @ W.double())
        dW = torch.empty(N, K, device="cuda"); dbv = torch.empty(N, device="cuda")
        ck(L.fc_k_dw(3, P(dev(G)), P(dev(A)), P(dW), P(dbv), M, N, K, S()))
        fig(f"linear dW  {name}", dW, G.t() @ A, G.double().t() @ A.double())
    # attention
    B = M // N_tok; d = 64
    qkv = torch.randn(B, N_tok, 3 * D, generator=g); dout = torch.randn(B, N_tok, D, generator=g) * 0.1

    def att(qkv_, dout_):
        q5 = qkv_.reshape(B, N_tok, 3, H, d).permute(2, 0, 3, 1, 4)
        q5 = q5.detach().requires_grad_(True)
        o = torch.nn.functional.softmax(q5[0] @ q5[1].transpose(-2, -1) * d ** -0.5, -1) @ q5[2]
        o2 = o.transpose(1, 2).reshape(B, N_tok, D)
        o2.backward(dout_)
        return o2.detach(), q5.grad.permute(1, 3, 0, 2, 4).reshape(B, N_tok, 3 * D)
    o64, d64 = att(qkv.double(), dout.double()); o32, d32 = att(qkv, dout)
    qd = dev(qkv); o = torch.zeros(B, N_tok, D, device="cuda"); lse = torch.zeros(B, H, N_tok, device="cuda"); delta = torch.zeros(B, H, N_tok, device="cuda")
    ck(L.fc_k_attention_fwd(1, 0, P(qd), P(o), P(lse), B, N_tok, H, d, d ** -0.5, S()))
    fig("attention fwd", o, o32, o64)
    dqkv = torch.zeros(B, N_tok, 3 * D, device="cuda")
    ck(L.fc_k_attention_bwd(1, 0, P(qd), P(o), P(dev(dout)), P(lse), P(delta), P(dqkv), B, N_tok, H, d, d ** -0.5, S()))
    fig("attention bwd", dqkv, d32, d64)
