"""Kernel-level parity on the GPU: each HIP kernel (called through the C ABI) against the oracle's formulas."""
import ctypes as C
import math

import pytest
import torch

from oracle import mome_oracle as O

pytestmark = pytest.mark.gpu


def L():
    from fedcola_amd import _lib
    return _lib


_KEEP = []


def dev(t, dt=None):
    """Device copy that stays alive until the end of the test (raw pointers are handed to the C ABI)."""
    t = t.cuda()
    t = t.to(dt) if dt is not None else t
    _KEEP.append(t)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return t


def maxerr(a, b):
    return float((a.detach().double().cpu() - b.double()).abs().max())


def amax(b):
    return max(1.0, float(b.abs().max()))


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


DT = {"fp32": (0, torch.float32, 2e-5), "bf16": (1, torch.bfloat16, 2e-2)}


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("M,D", [(5, 4), (37, 192), (300, 384), (64, 1024)])
def test_layernorm(prec, M, D):
    code, tdt, tol = DT[prec]
    g = torch.Generator().manual_seed(M * 1000 + D)
    x = torch.randn(M, D, generator=g).to(tdt).float()
    dy = torch.randn(M, D, generator=g).to(tdt).float()
    res = torch.randn(M, D, generator=g).to(tdt).float()
    gam = torch.randn(D, generator=g) * 0.2 + 1
    bet = torch.randn(D, generator=g) * 0.1
    y_ref, saved = O.ln_fwd(x, gam, bet, 1e-5)
    dx_ref, dg_ref, db_ref = O.ln_bwd(dy, gam, saved)
    xd, dyd, resd = dev(x, tdt), dev(dy, tdt), dev(res, tdt)
    y = torch.empty_like(xd); dx = torch.empty_like(xd)
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    lib = L().lib()
    L().check(lib.fc_k_layernorm_fwd(code, P(xd), P(dev(gam)), P(dev(bet)), P(y), P(mean), P(rstd), M, D, 1e-5, S()))
    L().check(lib.fc_k_layernorm_bwd(code, P(dyd), P(xd), P(mean), P(rstd), P(dev(gam)), P(resd), P(dx), P(dg), P(db), M, D, S()))
    torch.cuda.synchronize()
    e = maxerr(y, y_ref); assert e <= tol * amax(y_ref), f"y err {e}"
    e = maxerr(dx, dx_ref + res); assert e <= tol * amax(dx_ref + res), f"dx err {e}"
    e = maxerr(dg, dg_ref); assert e <= 1e-4 * amax(dg_ref), f"dg err {e}"
    e = maxerr(db, db_ref); assert e <= 1e-4 * amax(db_ref), f"db err {e}"
    # the in-model form: per-block partial rows + grouped reduction (no atomics on the hot path)
    dx2 = torch.empty_like(xd); dg2 = torch.zeros(D, device="cuda"); db2 = torch.zeros(D, device="cuda")
    part = torch.empty(int(lib.fc_k_layernorm_partial_floats(M, D)), device="cuda")
    L().check(lib.fc_k_layernorm_bwd_partial(code, P(dyd), P(xd), P(mean), P(rstd), P(dev(gam)), P(resd), P(dx2), P(dg2), P(db2), M, D, P(part), S()))
    torch.cuda.synchronize()
    e = maxerr(dx2, dx_ref + res); assert e <= tol * amax(dx_ref + res), f"dx (partial form) err {e}"
    e = maxerr(dg2, dg_ref); assert e <= 1e-4 * amax(dg_ref), f"dg (partial form) err {e}"
    e = maxerr(db2, db_ref); assert e <= 1e-4 * amax(db_ref), f"db (partial form) err {e}"


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(5, 7, 3), (70, 130, 33), (197 * 4, 384, 384), (256, 1152, 384), (394, 384, 1536), (1000, 192, 768),
                                   (197 * 22, 1152, 384), (4100, 1048, 200)])   # several persistent rounds per workgroup
def test_gemm(impl, kind, prec, M, N, K):
    # impl 1 with fp32 operands = the fp32 mode's GEMM: three bf16 MFMA products of split operands (fc_gemm_x3.hip), held to the same 1e-5 as
    # the VALU kernel; it declines shapes without 16-byte accesses (rc 1)
    code, tdt, tol = DT[prec]
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + kind)
    shpA = (M, K) if kind != 2 else (K, M)
    shpB = (N, K) if kind == 0 else (K, N)
    A = (torch.randn(*shpA, generator=g) * 0.5).to(tdt)
    Bm = (torch.randn(*shpB, generator=g) * 0.5).to(tdt)
    bias = torch.randn(N, generator=g)
    a = A.float() if kind != 2 else A.float().t()
    b = Bm.float().t() if kind == 0 else Bm.float()
    ref = a.double() @ b.double() + bias.double()
    for out_code, out_dt in ((code, tdt), (0, torch.float32)):
        Cd = torch.zeros(M, N, device="cuda", dtype=out_dt)
        rc = L().lib().fc_k_gemm(impl, kind, code, out_code, P(dev(A)), P(dev(Bm)), P(Cd), M, N, K, P(dev(bias)), 0, S())
        if rc == 1:
            pytest.skip("shape not covered by the MFMA fast path")
        L().check(rc)
        torch.cuda.synchronize()
        err = (Cd.double().cpu() - ref).abs().max().item()
        scale = ref.abs().max().item()
        t = 1e-5 if (prec == "fp32") else (1e-2 if out_dt == torch.bfloat16 else 2e-5 * math.sqrt(K) + 1e-6)
        assert err <= t * max(1.0, scale), (err, scale, out_dt)


@pytest.mark.parametrize("wide", [0, 1, 2])
@pytest.mark.parametrize("rows,out,inn", [(197 * 4, 384, 384), (1000, 1152, 384), (64 * 5 + 17, 1536, 384), (2048, 384, 1536), (777, 200, 768),
                                          (12608, 384, 384)])
def test_grouped_weight_gradient_kernels(wide, rows, out, inn):
    """dW = dY^T . X (fp32 out) and db = colsum(dY) through the 128x128 and the 128x384 grouped kernels: ragged row counts (k tail inside
    the last 64-row tile), out not a multiple of 128, two column tiles (in = 768 / 1536), the model's full reduction length."""
    g = torch.Generator().manual_seed(rows + out + inn)
    dY = (torch.randn(rows, out, generator=g) * 0.5).bfloat16()
    X = (torch.randn(rows, inn, generator=g) * 0.5).bfloat16()
    dW = torch.full((out, inn), float("nan"), device="cuda")
    db = torch.full((out,), float("nan"), device="cuda")
    L().check(L().lib().fc_k_dw(wide, P(dev(dY)), P(dev(X)), P(dW), P(db), rows, out, inn, S()))
    ref = dY.double().t() @ X.double()
    tol = 2e-5 * math.sqrt(rows) + 1e-6
    assert (dW.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    refb = dY.double().sum(0)
    assert (db.double().cpu() - refb).abs().max().item() <= tol * max(1.0, refb.abs().max().item())


@pytest.mark.parametrize("rows,out,inn", [(197 * 4, 384, 384), (777, 200, 768), (64 * 5 + 17, 1536, 384), (12608, 384, 1152), (12608, 1536, 384), (30, 8, 8)])
def test_fp32_mode_weight_gradient_kernel(rows, out, inn):
    """The fp32 mode's dW = dY^T . X and db += colsum(dY) (fc_gemm_x3.hip: three-way split operands, six bf16 MFMA products, the row
    reduction cut into slices whose partial tiles are added in fp64): ragged row counts, out / in not multiples of 128, the model's full reduction
    length -- held an order of magnitude tighter than an fp32 FMA chain would be (2e-6 of the largest element at 12 608 rows), the bias
    gradient accumulating onto what the buffer held, and the same bits on a second run (no atomics)."""
    g = torch.Generator().manual_seed(rows + 3 * out + 5 * inn)
    dY = torch.randn(rows, out, generator=g) * 0.5
    X = torch.randn(rows, inn, generator=g) * 0.5 + 0.1
    db0 = torch.randn(out, generator=g)
    res = []
    for _ in range(2):
        dW = torch.full((out, inn), float("nan"), device="cuda")
        db = dev(db0.clone())
        L().check(L().lib().fc_k_dw(3, P(dev(dY)), P(dev(X)), P(dW), P(db), rows, out, inn, S()))
        res.append((dW.cpu(), db.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ref = dY.double().t() @ X.double()
    assert (res[0][0].double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    refb = db0.double() + dY.double().sum(0)
    assert (res[0][1].double() - refb).abs().max().item() <= 2e-6 * max(1.0, refb.abs().max().item())


def attn_ref(qkv, B, N, H, d):
    q5 = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4).double()
    q, k, v = q5[0] * d ** -0.5, q5[1], q5[2]
    S_ = q @ k.transpose(-2, -1)
    Pm = torch.softmax(S_, -1)
    o = (Pm @ v).transpose(1, 2).reshape(B, N, H * d)
    lse = torch.logsumexp(S_, -1)
    return q, k, v, Pm, o, lse


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
# 193 / 208: first and last sequence length of the 13-valid-key-block forward specialisation (12 full blocks + 1 .. 16 keys), 209 / 224:
# the generic 14-block form beside it, 256: the 16-block form at its limit, 145: a 192-pixel image
@pytest.mark.parametrize("B,N,H,d", [(2, 5, 2, 2), (3, 197, 3, 64), (2, 32, 6, 64), (2, 16, 1, 64), (1, 40, 2, 32), (2, 70, 2, 64),
                                     (2, 193, 2, 64), (1, 208, 2, 64), (1, 209, 2, 64), (1, 224, 1, 64), (1, 256, 1, 64), (2, 145, 2, 64)])
def test_attention(impl, prec, B, N, H, d):
    # impl 1 with fp32 tensors = the fp32 mode's attention: v_mfma_f32_16x16x4_f32 chains (fc_attn_f32.hip), held to the VALU kernel's bounds;
    # it declines head dims other than 64 and sequences over 224 (rc 1)
    code, tdt, tol = DT[prec]
    g = torch.Generator().manual_seed(B * 100 + N)
    D = H * d
    qkv = (torch.randn(B, N, 3 * D, generator=g) * 1.0).to(tdt)
    dout = (torch.randn(B, N, D, generator=g) * 0.5).to(tdt)
    q, k, v, Pm, o_ref, lse_ref = attn_ref(qkv.float(), B, N, H, d)
    qd = dev(qkv); o = torch.zeros(B, N, D, device="cuda", dtype=tdt)
    lse = torch.zeros(B, H, N, device="cuda")
    scale = d ** -0.5
    rc = L().lib().fc_k_attention_fwd(impl, code, P(qd), P(o), P(lse), B, N, H, d, scale, S())
    if rc == 1:
        pytest.skip("shape not covered by the MFMA fast path")
    L().check(rc)
    torch.cuda.synchronize()
    t = 2e-5 if prec == "fp32" else 2e-2
    e = maxerr(o, o_ref); assert e <= t * amax(o_ref), f"o err {e}"
    e = maxerr(lse, lse_ref); assert e <= (1e-4 if prec == "fp32" else 2e-2), f"lse err {e}"
    # backward (uses the device's own o / lse)
    dO4 = dout.double().reshape(B, N, H, d).transpose(1, 2)
    dV = Pm.transpose(-2, -1) @ dO4
    dP = dO4 @ v.transpose(-2, -1)
    dS = Pm * (dP - (dP * Pm).sum(-1, keepdim=True))
    dq = (dS @ k) * scale
    dk = dS.transpose(-2, -1) @ q
    dref = torch.stack([dq, dk, dV], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * D)
    dqkv = torch.zeros(B, N, 3 * D, device="cuda", dtype=tdt)
    delta = torch.zeros(B, H, N, device="cuda")
    L().check(L().lib().fc_k_attention_bwd(impl, code, P(qd), P(o), P(dev(dout)), P(lse), P(delta), P(dqkv), B, N, H, d, scale, S()))
    torch.cuda.synchronize()
    err = (dqkv.double().cpu() - dref).abs().max().item()
    assert err <= (5e-5 if prec == "fp32" else 3e-2) * max(1, dref.abs().max().item()), err


def test_adamw_matches_oracle():
    n = 4096 + 64
    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g); gr = torch.randn(n, generator=g) * 1e-3
    m = torch.zeros(n); v = torch.zeros(n)
    pd, gd, md, vd = dev(p.clone()), dev(gr.clone()), dev(m.clone()), dev(v.clone())
    for step in (1, 2, 3):
        O.adamw_step(p, gr, m, v, step, 1e-3, weight_decay=0.01)
        L().check(L().lib().fc_k_adamw(P(pd), P(gd), P(md), P(vd), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, S()))
    torch.cuda.synchronize()
    assert maxerr(pd, p) <= 2e-6, maxerr(pd, p)
    assert maxerr(md, m) <= 1e-9 and maxerr(vd, v) <= 1e-12


def test_cast_bf16_rne():
    x = torch.randn(1000) * 3
    y = torch.empty(1000, device="cuda", dtype=torch.bfloat16)
    L().check(L().lib().fc_k_cast(1, P(dev(x)), P(y), 1000, S()))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))


def _bf(t):
    return t.to(torch.bfloat16).double()


@pytest.mark.probes
@pytest.mark.parametrize("M", [64, 197, 333, 197 * 22, 2048])      # ragged last panel, one full-size chain, the text tower
@pytest.mark.parametrize("Hd", [1536, 256])
@pytest.mark.parametrize("rowscale", [False, True])
def test_mlp_fused(M, Hd, rowscale):
    """Fused MLP (fc_mlp.hip): Mlp.forward mome.py:117-123 + the residual of Block.forward mome.py:228, and its backward mirror, against
    (a) the exact formulas in fp64 with bf16 rounding at the kernel's store points and (b) the separate GEMM kernels it replaces."""
    if rowscale and Hd == 256:
        pytest.skip("one drop-path case is enough")
    D, N_tok = 384, 197 if M % 197 == 0 else 32
    g = torch.Generator().manual_seed(M + Hd)
    X = (torch.randn(M, D, generator=g)).to(torch.bfloat16)
    W1 = (torch.randn(Hd, D, generator=g) * D ** -0.5).to(torch.bfloat16)
    W2 = (torch.randn(D, Hd, generator=g) * Hd ** -0.5).to(torch.bfloat16)
    b1 = torch.randn(Hd, generator=g) * 0.2
    b2 = torch.randn(D, generator=g) * 0.2
    res = torch.randn(M, D, generator=g).to(torch.bfloat16)
    dm = torch.randn(M, D, generator=g).to(torch.bfloat16)
    nsamp = (M + N_tok - 1) // N_tok
    rs = (torch.rand(nsamp, generator=g) > 0.3).float() / 0.7 if rowscale else None
    lib = L().lib()
    Xd, W1d, W2d, b1d, b2d, resd, dmd = dev(X), dev(W1), dev(W2), dev(b1), dev(b2), dev(res), dev(dm)
    rsd = dev(rs) if rowscale else None
    act = torch.full((M, Hd), 7.0, device="cuda", dtype=torch.bfloat16); gs = torch.full_like(act, 7.0)
    out = torch.full((M, D), 7.0, device="cuda", dtype=torch.bfloat16)
    pf = torch.empty(2 * D * Hd, device="cuda", dtype=torch.bfloat16); pb = torch.empty_like(pf)
    L().check(lib.fc_k_mlp_pack(P(W1d), P(W2d), P(pf), P(pb), D, Hd, S()))
    rc = lib.fc_k_mlp_fused(0, P(Xd), P(pf), P(b1d), P(b2d), P(act), P(gs), P(resd), P(rsd), N_tok, P(out), M, D, Hd, S())
    L().check(rc)
    torch.cuda.synchronize()
    # (a) exact formulas
    u = X.double() @ W1.double().t() + b1.double()
    cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2))); pdf = torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
    h_ref, gp_ref = u * cdf, cdf + u * pdf
    y = _bf(h_ref) @ W2.double().t() + b2.double()
    if rowscale:
        y = y * rs.double().repeat_interleave(N_tok)[:M, None]
    o_ref = y + res.double()
    e = maxerr(act, h_ref); assert e <= 1e-2 * amax(h_ref), f"gelu(u) err {e}"
    e = maxerr(gs, gp_ref); assert e <= 1e-2 * amax(gp_ref), f"gelu'(u) err {e}"
    e = maxerr(out, o_ref); assert e <= 1.2e-2 * amax(o_ref), f"out err {e}"
    # (b) the separate kernels: fc1 with the GELU epilogue, fc2 with bias + residual (no drop-path form in that entry point)
    act2 = torch.empty_like(act); gs2 = torch.empty_like(gs); out2 = torch.empty_like(out)
    L().check(lib.fc_k_gemm_epi(0, P(Xd), P(W1d), P(act2), M, Hd, D, P(b1d), None, P(gs2), None, S()))
    torch.cuda.synchronize()
    assert torch.equal(act, act2), f"gelu(u): {int((act != act2).sum())} elements differ from the separate kernel"
    assert torch.equal(gs, gs2), f"gelu'(u): {int((gs != gs2).sum())} elements differ from the separate kernel"
    if not rowscale:
        L().check(lib.fc_k_gemm_epi(0, P(act2), P(W2d), P(out2), M, D, Hd, P(b2d), P(resd), None, None, S()))
        torch.cuda.synchronize()
        nd = int((out != out2).sum())      # (the residual add may contract differently: a rare one-ulp flip is allowed, nothing more)
        assert nd <= 1e-3 * out.numel() and maxerr(out, out2.double().cpu()) <= 2e-2 * amax(o_ref), f"out: {nd} elements differ from the separate kernels"
    # ---- backward: du = (dm . W2) * gelu'(u), dx = du . W1
    du = torch.full((M, Hd), 7.0, device="cuda", dtype=torch.bfloat16); dx = torch.full((M, D), 7.0, device="cuda", dtype=torch.bfloat16)
    L().check(lib.fc_k_mlp_fused(1, P(dmd), P(pb), None, None, P(du), P(gs), None, None, 1, P(dx), M, D, Hd, S()))
    torch.cuda.synchronize()
    du_ref = (dm.double() @ W2.double()) * gs.double().cpu()
    dx_ref = _bf(du_ref) @ W1.double()
    e = maxerr(du, du_ref); assert e <= 1e-2 * amax(du_ref), f"du err {e}"
    e = maxerr(dx, dx_ref); assert e <= 1.2e-2 * amax(dx_ref), f"dx err {e}"
    du2 = torch.empty_like(du); dx2 = torch.empty_like(dx)
    L().check(lib.fc_k_gemm_epi(1, P(dmd), P(W2d), P(du2), M, Hd, D, None, None, None, P(gs), S()))
    L().check(lib.fc_k_gemm_epi(1, P(du2), P(W1d), P(dx2), M, D, Hd, None, None, None, None, S()))
    torch.cuda.synchronize()
    assert torch.equal(du, du2), f"du: {int((du != du2).sum())} elements differ from the separate kernel"
    assert torch.equal(dx, dx2), f"dx: {int((dx != dx2).sum())} elements differ from the separate kernels"
