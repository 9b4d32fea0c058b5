"""Known-answer tests of the contrastive loss.

The reference takes ``ContrastiveLossWithTemperature`` from torchmultimodal (src/criterions/__init__.py:3,8; call site
src/client/fedavgclient.py:95), a third-party package that is neither vendored nor pinned nor installed: parity is UNPINNED at
that boundary.  Its published definition -- logit_scale = ln(1/0.07) clamped to [0, ln 100], L = exp(logit_scale) * A B^T,
loss = (CE(L, arange) + CE(L^T, arange)) / 2 with mean reduction -- is pinned here by answers that do not come from the oracle:
hand arithmetic for B = 2 and closed forms for structured inputs, for the oracle (CPU) and for the HIP kernels (GPU)."""
import math

import pytest
import torch

from oracle import mome_oracle as O

TAU = 1.0 / 0.07


def test_temperature_is_one_over_0p07_in_fp32():
    # exp(clamp(ln(1/0.07), 0, ln 100)) evaluated in fp32, as an nn.Parameter would hold it
    t = O.contrastive_tau()
    assert abs(t - TAU) <= 2e-6 * TAU
    assert t == float(torch.exp(torch.tensor(math.log(1 / 0.07), dtype=torch.float32)))


def _hand_b2(a, b, tau):
    """B = 2 by hand with Python floats: logits, the two cross entropies, their mean."""
    L = [[tau * sum(x * y for x, y in zip(a[i], b[j])) for j in range(2)] for i in range(2)]
    def lse(u, v):
        m = max(u, v)
        return m + math.log(math.exp(u - m) + math.exp(v - m))
    rows = [lse(L[i][0], L[i][1]) - L[i][i] for i in range(2)]
    cols = [lse(L[0][j], L[1][j]) - L[j][j] for j in range(2)]
    return 0.5 * (sum(rows) / 2 + sum(cols) / 2), L


def test_b2_hand_computed():
    a = [[0.6, 0.8], [1.0, 0.0]]
    b = [[0.8, 0.6], [0.0, 1.0]]
    exp, L = _hand_b2(a, b, TAU)
    # the numbers, spelled out: L = tau * [[0.96, 0.8], [0.8, 0.0]]
    assert abs(L[0][0] - 0.96 * TAU) < 1e-12 and abs(L[1][1]) < 1e-12
    row0 = math.log(1 + math.exp((0.8 - 0.96) * TAU))
    row1 = math.log(1 + math.exp(0.8 * TAU))
    col0 = math.log(1 + math.exp((0.8 - 0.96) * TAU))
    col1 = math.log(1 + math.exp(0.8 * TAU))
    assert abs(exp - 0.5 * ((row0 + row1) / 2 + (col0 + col1) / 2)) < 1e-12
    loss, da, db = O.contrastive_loss(torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64), tau=TAU)
    assert abs(float(loss) - exp) < 1e-12


@pytest.mark.parametrize("B", [2, 5, 64])
def test_closed_forms(B):
    D = max(B, 8)
    eye = torch.eye(B, D, dtype=torch.float64)
    # identical orthonormal embeddings: L = tau*I  ->  loss = ln(1 + (B-1) e^-tau)
    loss, _, _ = O.contrastive_loss(eye, eye, tau=TAU)
    assert abs(float(loss) - math.log1p((B - 1) * math.exp(-TAU))) < 1e-12
    # every embedding equal: L = tau everywhere  ->  loss = ln B
    u = torch.zeros(B, D, dtype=torch.float64); u[:, 0] = 1.0
    loss, da, db = O.contrastive_loss(u, u, tau=TAU)
    assert abs(float(loss) - math.log(B)) < 1e-12
    # ... and the gradient of a row: tau * sum_j dL_ij b_j with dL_ij = (1/B - delta_ij)/B (both halves equal) -> zero here
    assert float(da.abs().max()) < 1e-12 and float(db.abs().max()) < 1e-12
    # anti-aligned orthonormal pairs: L = -tau*I  ->  loss = tau + ln((B-1) + e^-tau)
    loss, _, _ = O.contrastive_loss(eye, -eye, tau=TAU)
    assert abs(float(loss) - (TAU + math.log((B - 1) + math.exp(-TAU)))) < 1e-10
    # a permutation of the partners (a_i pairs with b_{i+1}): the diagonal is 0, one off-diagonal entry per row / column is tau
    perm = torch.roll(eye, 1, 0)
    loss, _, _ = O.contrastive_loss(eye, perm, tau=TAU)
    exp = math.log(math.exp(TAU) + (B - 1)) if B > 1 else 0.0       # lse - L_ii with L_ii = 0
    assert abs(float(loss) - exp) < 1e-10


def test_gradients_match_autograd_of_the_published_formula():
    g = torch.Generator().manual_seed(4)
    a = torch.nn.functional.normalize(torch.randn(7, 16, generator=g, dtype=torch.float64), dim=-1).requires_grad_()
    b = torch.nn.functional.normalize(torch.randn(7, 16, generator=g, dtype=torch.float64), dim=-1).requires_grad_()
    L = TAU * a @ b.t()
    lab = torch.arange(7)
    ref = 0.5 * (torch.nn.functional.cross_entropy(L, lab) + torch.nn.functional.cross_entropy(L.t(), lab))
    ref.backward()
    loss, da, db = O.contrastive_loss(a.detach(), b.detach(), tau=TAU)
    assert abs(float(loss) - float(ref)) < 1e-12
    assert float((da - a.grad).abs().max()) < 1e-12 and float((db - b.grad).abs().max()) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("B", [2, 5, 64])
def test_hip_kernels_closed_forms(B):
    """fc_contrastive_loss_fwd_bwd (k_con_logits / k_con_lse / k_con_grads) against the same closed forms."""
    from fedcola_amd import _lib
    L = _lib.lib(); P = _lib.ptr
    D = max(B, 8)
    tau = O.contrastive_tau()

    def run(a, b):
        a, b = a.float().cuda().contiguous(), b.float().cuda().contiguous()
        scratch = torch.empty(int(L.fc_contrastive_scratch_floats(B)), device="cuda")
        lossbuf = torch.zeros(2, device="cuda"); da = torch.empty_like(a); db = torch.empty_like(b)
        _lib.check(L.fc_contrastive_loss_fwd_bwd(P(a), P(b), B, D, tau, P(scratch), scratch.numel(), P(lossbuf), P(da), P(db), _lib.stream_ptr()))
        torch.cuda.synchronize()
        return float(lossbuf[1]), float(lossbuf[0]), da.cpu(), db.cpu()

    eye = torch.eye(B, D)
    loss, lsum, _, _ = run(eye, eye)
    assert abs(loss - math.log1p((B - 1) * math.exp(-tau))) < 1e-6 and abs(lsum - loss * B) < 1e-5 * max(1.0, abs(loss * B))
    u = torch.zeros(B, D); u[:, 0] = 1.0
    loss, _, da, db = run(u, u)
    assert abs(loss - math.log(B)) < 1e-5 and float(da.abs().max()) < 1e-5 and float(db.abs().max()) < 1e-5
    loss, _, _, _ = run(eye, -eye)
    assert abs(loss - (tau + math.log((B - 1) + math.exp(-tau)))) < 1e-4
    # B = 2 hand-computed
    if B == 2:
        a = [[0.6, 0.8] + [0.0] * (D - 2), [1.0, 0.0] + [0.0] * (D - 2)]
        b = [[0.8, 0.6] + [0.0] * (D - 2), [0.0, 1.0] + [0.0] * (D - 2)]
        exp, _ = _hand_b2(a, b, tau)
        loss, _, da, db = run(torch.tensor(a), torch.tensor(b))
        assert abs(loss - exp) < 1e-5 * max(1.0, exp)
        lo, dao, dbo = O.contrastive_loss(torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64), tau=tau)
        assert float((da.double() - dao).abs().max()) < 1e-5 and float((db.double() - dbo).abs().max()) < 1e-5


def test_autograd_form_of_the_oracle_step_equals_the_explicit_backward():
    """bench.py's cpu_baseline times the faster of the two forms of the oracle step: explicit backward (client_step) and
    torch.autograd over the same forward (client_step_autograd, what the reference's loss.backward() does).  Same loss, gradients and
    updated weights."""
    import golden_util as G
    from test_oracle_golden import cfg_from_mk
    rec = G.load("model_small.json")
    cfg = cfg_from_mk(rec["mk"])
    img, ids, _ = G.case_inputs(rec)
    p1, p2 = G.case_weights("small"), G.case_weights("small")
    l1, _, g1 = O.client_step(p1, cfg, ("img+txt", img, ids), dict(step=0, m={}, v={}), lr=1e-4)
    l2, _, g2 = O.client_step_autograd(p2, cfg, ("img+txt", img, ids), dict(step=0, m={}, v={}), lr=1e-4)
    assert abs(float(l1) - float(l2)) <= 1e-6
    assert set(g1) == set(g2)
    for k in g1:
        assert float((g1[k] - g2[k]).abs().max()) <= 2e-5 * max(float(g1[k].abs().max()), 1e-6), k
