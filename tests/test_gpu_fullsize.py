"""Full-size parity on the GPU: mome_small_patch16 (ViT-S + 12x384 text tower, vocab 7732, 32-token captions), B=64 --
BASELINE.json's config[1] -- one client step against the oracle on the same synthetic batch.
fp32 mode: <= 1e-4 relative on features, loss and every gradient tensor.  The bf16 (timed) mode is held to the emulating oracle in
tests/test_gpu_bf16_parity.py (layer-local 1e-2 relative L2 on every tensor, end-to-end statistical bound)."""
import pytest
import torch

import product_util as PU
from oracle import mome_oracle as O

pytestmark = pytest.mark.gpu

MK = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=384, depth=12, num_heads=6,
          vocab_size=7732, max_text_len=32)


def _batch(B=64, seq=32, vocab=7732):
    g = torch.Generator().manual_seed(1000)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
    ids = torch.randint(1, vocab, (B, seq), generator=g)
    lens = torch.randint(8, seq + 1, (B,), generator=g)
    ids[torch.arange(seq)[None, :] >= lens[:, None]] = 0
    return img, ids


@pytest.fixture(scope="module")
def oracle_result():
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    torch.manual_seed(5)
    ref = M(**MK)                                   # reference default init (pos/cls zero) ...
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    for k in sd:                                    # ... with non-zero pos / cls so that their gradients are exercised
        if "pos_embed" in k or "cls_token" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    img, ids = _batch()
    cfg = O.OracleCfg(D=384, depth=12, heads=6, vocab=7732, max_text_len=32)
    p = {k: v.clone() for k, v in sd.items()}
    outs, cache = O.forward(p, cfg, [img, ids], feat_out=True)
    loss, da, db = O.contrastive_loss(outs[0], outs[1])
    grads = O.backward(p, cfg, cache, [da, db])
    return sd, img, ids, outs, float(loss), grads


@pytest.mark.parametrize("prec,otol,gtol", [("fp32", 1e-4, 1e-4)])
def test_vit_s_b64_step_vs_oracle(oracle_result, prec, otol, gtol):
    sd, img, ids, outs_o, loss_o, grads_o = oracle_result
    model = PU.build_product(MK, prec, sd)
    model.train()
    with torch.no_grad():
        outs = model([img.cuda(), ids.cuda()], feat_out=True)
    for o, oo in zip(outs, outs_o):
        assert float((o.cpu() - oo).abs().max()) <= otol, "encoder outputs (unit-norm features)"
    loss, grads, _ = PU.product_step(model, "img+txt", img, ids, None, 1e-4)
    assert abs(loss - loss_o) <= max(otol, 1e-4) * max(1.0, abs(loss_o))
    worst = ("", 0.0)
    for k, go in grads_o.items():
        gk = grads[k]
        if k.endswith("attn.qkv.bias"):      # the key third has an exactly-zero true gradient (softmax shift invariance): rounding noise on both sides
            D3 = go.numel() // 3
            sel = torch.cat([torch.arange(0, D3), torch.arange(2 * D3, 3 * D3)])
            gk, go = gk[sel], go[sel]
        scale = max(float(go.abs().max()), 1e-7)
        rel = float((gk - go).abs().max()) / scale
        if rel > worst[1]:
            worst = (k, rel)
    print(f"ViT-S B=64 {prec}: worst gradient tensor {worst}")
    assert worst[1] <= gtol, f"worst gradient tensor {worst}"


# ---- BASELINE.json config[3]: the 768-wide (ViT-B/16 + BERT-base-width) model, here 2 layers deep so that the oracle finishes in
# seconds; one img+txt step and one uni-modal step with trained aux (--aux --aux_trained) against the oracle.
MKB = dict(embed_dim=768, depth=2, num_heads=12, vocab_size=30522, max_text_len=40)


ETA_BOUND = 1e-4      # per unit of condition number: north_star's own tolerance for a tensor that does not cancel (kappa = 1)


@pytest.mark.parametrize("kind,prec,otol,gtol", [("img+txt", "fp32", 1e-4, 1e-4), ("img", "fp32", 1e-4, 1e-4)])
def test_vit_b_width_step_vs_oracle(kind, prec, otol, gtol):
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    from synth import det_state_dict
    if kind == "img+txt":
        mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], **MKB)
        cfg = O.OracleCfg(D=768, depth=2, heads=12, vocab=30522, max_text_len=40)
    else:
        mk = dict(modalities=["img", None], num_classes=[100, None], tasks=["cls", None], with_aux=True, aux_trained=True, **MKB)
        cfg = O.OracleCfg(modalities=("img", None), tasks=("cls", None), num_classes=(100, None), D=768, depth=2, heads=12, vocab=30522,
                          max_text_len=40, with_aux=True, aux_trained=True)
    torch.manual_seed(2)
    shapes = {k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}
    sd = det_state_dict(shapes, base_seed=41)
    B = 8
    img, ids = _batch(B, 40, 30522)
    y = (torch.arange(B) * 13 + 5) % 100
    p = {k: v.clone() for k, v in sd.items()}
    batch = ("img+txt", img, ids) if kind == "img+txt" else ("img", img, y)
    loss_o, outs_o, grads_o = O.client_step(p, cfg, batch, dict(step=0, m={}, v={}), lr=1e-4)
    model = PU.build_product(mk, prec, sd)
    model.train()
    loss, grads, _ = PU.product_step(model, kind, img, ids, y, 1e-4)
    assert abs(loss - float(loss_o)) <= max(otol, 1e-4) * max(1.0, abs(float(loss_o)))
    worst, worst1d = ("", 0.0), ("", 0.0)
    for k, go in grads_o.items():
        if "cross_modal_scale" in k:
            continue                                   # cancellation-dominated scalar, bounded separately in test_gpu_model
        gk = grads[k]
        if k.endswith("attn.qkv.bias"):      # the key third has an exactly-zero true gradient (softmax shift invariance): rounding noise on both sides
            D3 = go.numel() // 3
            sel = torch.cat([torch.arange(0, D3), torch.arange(2 * D3, 3 * D3)])
            gk, go = gk[sel], go[sel]
        scale = max(float(go.abs().max()), 1e-7)
        rel = float((gk - go).abs().max()) / scale
        gemm_like = k.endswith(("qkv.weight", "proj.weight", "fc1.weight", "fc2.weight", "aux_weight", "head.weight")) and "embeddings" not in k
        if gemm_like:                                   # the linears' weights: well-conditioned sums, held to gtol against the fp32 oracle
            worst = max(worst, (k, rel), key=lambda t: t[1])
        else:                                           # column sums: biases, LayerNorm vectors, embedding tables, cls / position rows
            worst1d = max(worst1d, (k, rel), key=lambda t: t[1])
    print(f"{kind} {prec}: worst linear weight gradient {worst}, worst column-sum gradient {worst1d} (against the fp32 oracle)")
    assert worst[1] <= gtol, f"worst gradient tensor {worst}"
    # Bias / LayerNorm vectors and the embedding tables' rows at B = 8 are column sums over a few rows to a few hundred rows (320 text rows)
    # that cancel to ~1e-3 of their summands.  The library forms the long sums in fp64 (k_colsum_f64, fp64 LayerNorm partial rows), so the
    # summation adds nothing -- but the SUMMANDS of two fp32 implementations differ by ~1e-6 each (the library's GEMMs are not torch's: since
    # round 4 they are split-operand MFMA products, 5x closer to the exact product than an fp32 FMA chain, fc_gemm_x3.hip), which after the
    # cancellation is ~1e-4 of the result's maximum: that is the conditioning of the quantity, not an error of either side.  So EVERY
    # gradient tensor is held to 1e-4 against the EXACT gradient (the same oracle run in fp64), and the fp32 oracle's own distance from it
    # is printed beside ours; against the fp32 oracle the bound for the column sums is the sum of the two (2e-4).
    p64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    b64 = ("img+txt", img.double(), ids) if kind == "img+txt" else ("img", img.double(), y)
    # While the exact gradient is computed, record the MASS of every sum a LayerNorm or a linear layer reduces over the rows:
    # sum_r |summand_r| per output element.  A gradient element computed in fp32 from fp32 activations cannot be closer
    # to the exact value than a few ulps of its summands, whatever the summation order or precision -- its honest error scale is the mass, not
    # its own (cancelled) size.
    masses = []
    ln_orig, lin_orig = O.ln_bwd, O.linear_bwd

    def ln_rec(dy, g, saved):
        dx, dg, db = ln_orig(dy, g, saved)
        D = saved[0].shape[-1]
        masses.append((dg, (dy * saved[0]).abs().reshape(-1, D).sum(0)))
        masses.append((db, dy.abs().reshape(-1, D).sum(0)))
        if dx.dim() == 3:      # the text embedding LayerNorm's dx is summed over the batch (position rows) and over every token (type row)
            masses.append((dx.sum(0), dx.abs().sum(0)))
            masses.append((dx.reshape(-1, D).sum(0), dx.abs().reshape(-1, D).sum(0)))
        return dx, dg, db

    blk_orig = O.block_bwd

    def blk_rec(p_, pre, *a, **kw):
        dh = blk_orig(p_, pre, *a, **kw)
        if pre.endswith(".0"):       # the first block's dx: summed over the batch into pos_embed, its row 0 into cls_token
            masses.append((dh.sum(0, keepdim=True), dh.abs().sum(0, keepdim=True)))
            masses.append((dh[:, 0].sum(0).reshape(1, 1, -1), dh[:, 0].abs().sum(0).reshape(1, 1, -1)))
        return dh

    def lin_rec(dy, x, W):
        dx, dW, db = lin_orig(dy, x, W)
        dy2, x2 = dy.reshape(-1, dy.shape[-1]).abs(), x.reshape(-1, x.shape[-1]).abs()
        masses.append((dW, dy2.t() @ x2))
        masses.append((db, dy2.sum(0)))
        return dx, dW, db
    O.ln_bwd, O.linear_bwd, O.block_bwd = ln_rec, lin_rec, blk_rec
    try:
        _, _, grads_64 = O.client_step(p64, cfg, b64, dict(step=0, m={}, v={}), lr=1e-4)
    finally:
        O.ln_bwd, O.linear_bwd, O.block_bwd = ln_orig, lin_orig, blk_orig
    mass_of = {}
    for k, g64 in grads_64.items():
        for val, mass in masses:
            if val.shape == g64.shape and torch.equal(val, g64):
                mass_of[k] = mass
                break
            # an embedding table whose leading rows (position rows 0 .. Nt - 1) or first row (token type 0) received the sum
            lead = g64[:val.shape[0]] if val.dim() == 2 and g64.dim() == 2 and val.shape[0] < g64.shape[0] else g64[0] if val.dim() == 1 and g64.dim() == 2 else None
            if lead is not None and lead.shape == val.shape and float(val.abs().max()) > 0 and torch.equal(val, lead) and not bool(g64[val.shape[0] if val.dim() == 2 else 1:].any()):
                m = torch.zeros_like(g64)
                m[:val.shape[0]] = mass if val.dim() == 2 else 0
                if val.dim() == 1:
                    m[0] = mass
                mass_of[k] = m
                break
    # Per tensor: rel = max |library - exact| / max |exact|, kappa = max mass / max |exact| (how much larger the summands are than what is
    # left of them: the condition number of the tensor as a sum), and rel / kappa = the error in units of the summands' mass.
    rows = []
    for k, g64 in grads_64.items():
        if "cross_modal_scale" in k:
            continue
        gk, go = grads[k].double(), grads_o[k].double()
        mass = mass_of.get(k)
        if k.endswith("attn.qkv.bias"):
            D3 = g64.numel() // 3
            sel = torch.cat([torch.arange(0, D3), torch.arange(2 * D3, 3 * D3)])
            gk, go, g64 = gk[sel], go[sel], g64[sel]
            mass = mass[sel] if mass is not None else None
        scale = max(float(g64.abs().max()), 1e-7)
        kappa = max(1.0, float(mass.max()) / scale) if mass is not None else 1.0
        rows.append((k, kappa, float((gk - g64).abs().max()) / scale, float((go - g64).abs().max()) / scale))
    rows.sort(key=lambda r: -r[2])
    print(f"{kind} {prec}: gradients against the EXACT (fp64) gradient, worst 8 of {len(rows)} tensors ({len(mass_of)} with recorded summands):")
    for k, kappa, mine, theirs in rows[:8]:
        print(f"    {k:44s} kappa {kappa:6.1f}  library {mine:.2e} ({mine / kappa:.2e} per unit kappa)  fp32 oracle {theirs:.2e} ({theirs / kappa:.2e})")
    eta_l = max(rows, key=lambda r: r[2] / r[1]); eta_o = max(rows, key=lambda r: r[3] / r[1])
    print(f"{kind} {prec}: worst error per unit of condition number: library {eta_l[2] / eta_l[1]:.2e} ({eta_l[0]}), fp32 oracle {eta_o[3] / eta_o[1]:.2e} ({eta_o[0]});"
          f" bound {ETA_BOUND:.1e}")
    if prec == "fp32":
        # The conditioning-scaled statement: every gradient tensor is within ETA_BOUND x kappa of the exact gradient,
        # relative to its maximum.  ETA_BOUND leaves >= 3x headroom over the measured worst (profiles/r05/parity_margins.txt) and is BELOW
        # north_star's 1e-4 for a tensor that does not cancel (kappa = 1); a tensor whose column sums cancel to 1 / kappa of their summands
        # gets kappa times that -- the fp32 oracle's own distance from the exact gradient, printed beside ours, scales the same way.
        assert len(mass_of) >= 0.7 * len(rows), "the mass recorder lost track of the block gradients"
        for k, kappa, mine, theirs in rows:
            assert mine <= ETA_BOUND * kappa, f"{k}: {mine:.2e} of its maximum from the exact gradient at condition number {kappa:.1f} (bound {ETA_BOUND * kappa:.2e})"
        # ... AND north_star's plain bound, without the conditioning: every tensor within 1e-4 of the exact gradient relative to its
        # maximum (the kappa form alone would let a 10x regression of a cancelling tensor through).  The worst tensor
        # (blockses.1.1.norm1.bias, kappa 9.3) measures 9.62e-5: a thin margin, but the fp32 mode has no atomics on these tensors and a fixed
        # reduction order -- the figure is the same on every run and every box until a kernel changes, and then this line is the alarm.
        for k, kappa, mine, theirs in rows:
            assert mine <= 1e-4, f"{k}: {mine:.2e} of its maximum from the exact gradient (plain bound 1e-4; condition number {kappa:.1f})"
    else:
        assert worst1d[1] <= gtol, f"worst 1-D gradient tensor {worst1d}"
