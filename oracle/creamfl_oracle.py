"""CPU restatement of the reference's CreamFL path.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and smoke / cpu_baseline legs) as the checker of the HIP path, never by the product.

Follows /root/reference/src/client/creamflclient.py and /root/reference/src/server/creamflserver.py:
  moon_ce, inter_ce          creamflclient.py:160-186 (uni-modal), :196-225 (img+txt): temperature 0.5, nn.CrossEntropyLoss (mean)
  clip_grad_norm             torch.nn.utils.clip_grad_norm_(params, 2) as called at creamflclient.py:232, creamflserver.py:334
  client_update              CreamflClient.update :70-247: local epoch(s) with the task loss, then one pass over the public set
                             per epoch with (loss_moon|intra + loss_inter) * interintra_weight, clip 2, the SAME optimizer
  pub_features               update_pub_feature :38-66
  aggregate_features         the `aggregation` closure of CreamflServer.update :372-407
  zero_init_aggregate        CreamflServer._aggregate :251-288 (weighted sum into zeros, plain coefficients)
  kd_distill                 CreamflServer._aggregate :294-336 (MSE to the aggregated public features, AdamW(p_lr), clip 2)
Pinned by tests/golden/cream.json (the real classes run on toy models, tests/golden/make_golden.py cream).
torch.optim.AdamW semantics kept: a parameter whose gradient is None in a step is skipped AND keeps its own step count."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from . import mome_oracle as O

Tensor = torch.Tensor
TEMP = 0.5


def _ce_rows(logits: Tensor, labels: Tensor):
    """mean cross-entropy and d/dlogits"""
    lse = torch.logsumexp(logits, dim=1, keepdim=True)
    logp = logits - lse
    n = logits.shape[0]
    loss = -logp[torch.arange(n), labels].mean()
    d = torch.exp(logp)
    d[torch.arange(n), labels] -= 1.0
    return loss, d / n


def moon_ce(f: Tensor, target: Tensor, old: Tensor, rows_norm: Optional[int] = None):
    """logits = [f.target, f.old] / 0.5, label 0 (creamflclient.py:175-186).  Returns (sum over rows / rows_norm, dL/df)."""
    pos, neg = (f * target).sum(-1), (f * old).sum(-1)
    z = torch.stack([pos, neg], dim=1) / TEMP
    n = rows_norm or f.shape[0]
    lse = torch.logsumexp(z, dim=1)
    loss = (lse - z[:, 0]).sum() / n
    p = torch.softmax(z, dim=1)
    dpos = (p[:, 0] - 1.0) / n / TEMP
    dneg = p[:, 1] / n / TEMP
    return loss, dpos[:, None] * target + dneg[:, None] * old


def inter_ce(f: Tensor, G: Tensor, labels: Tensor):
    """CE(f @ G.T / 0.5, labels) (creamflclient.py:165,173,181-182).  Returns (loss, dL/df)."""
    loss, dz = _ce_rows(f @ G.t() / TEMP, labels)
    return loss, dz @ G / TEMP


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float = 2.0) -> float:
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads.values()]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return float(total)


def adam_apply(p, grads, state, lr, weight_decay=0.0):
    """torch.optim.AdamW.step: per-parameter step counts, parameters without a gradient are skipped."""
    for k, g in grads.items():
        if k not in state["m"]:
            state["m"][k] = torch.zeros_like(p[k])
            state["v"][k] = torch.zeros_like(p[k])
            state["t"][k] = 0
        state["t"][k] += 1
        O.adamw_step(p[k], g, state["m"][k], state["v"][k], state["t"][k], lr, weight_decay=weight_decay)


def _features(p, cfg, kind, x_img, x_ids):
    outs, cache = O.forward(p, cfg, [x_img if kind != "txt" else None, x_ids if kind != "img" else None], feat_out=True)
    return outs, cache


def client_update(p, cfg, kind, train_batches, pub_batches, distill_index: Sequence[int], g_img: Tensor, g_txt: Tensor, E: int, lr: float,
                  interintra_weight: float, n_train: int, weight_decay: float = 0.0, no_mm_contrastive: bool = False):
    """CreamflClient.update.  kind in {'img','txt','img+txt'}; train_batches: list of oracle batches
    (('img', x, y) | ('txt', ids, y) | ('img+txt', img, ids)); pub_batches: list of (img, ids, index).  Mutates p.
    Returns {epoch: loss}."""
    old = {k: v.clone() for k, v in p.items()}
    dd = {int(b): a for a, b in enumerate(distill_index)}
    state = dict(m={}, v={}, t={})
    results = {}
    tkeys = set(O.trainable_keys(p, cfg))
    for e in range(E):
        tot = 0.0
        for batch in train_batches:
            if kind == "img+txt":
                outs, cache = O.forward(p, cfg, [batch[1], batch[2]], feat_out=True)
                loss, da, db = O.contrastive_loss(outs[0], outs[1])
                d_outs, nb = [da, db], batch[1].shape[0]
            elif kind == "img":
                outs, cache = O.forward(p, cfg, [batch[1], None])
                loss, dl = O.cross_entropy(outs[0], batch[2])
                d_outs, nb = [dl, None], batch[1].shape[0]
            else:
                outs, cache = O.forward(p, cfg, [None, batch[1]])
                loss, dl = O.cross_entropy(outs[1], batch[2])
                d_outs, nb = [None, dl], batch[1].shape[0]
            grads = {k: g for k, g in O.backward(p, cfg, cache, d_outs).items() if k in tkeys}
            adam_apply(p, grads, state, lr, weight_decay)
            tot += float(loss) * nb
        results[e + 1] = tot / n_train
        if interintra_weight > 0 and not (no_mm_contrastive and kind == "img+txt"):
            for img, ids, index in pub_batches:
                d_idx = torch.tensor([dd[int(i)] for i in index])
                outs, cache = _features(p, cfg, kind, img, ids)
                with torch.no_grad():
                    oouts, _ = _features(old, cfg, kind, img, ids)
                if kind == "img":
                    lm, d1 = moon_ce(outs[0], g_img[d_idx], oouts[0])
                    li, d2 = inter_ce(outs[0], g_txt, d_idx)
                    d_outs = [(d1 + d2) * interintra_weight, None]
                elif kind == "txt":
                    lm, d1 = moon_ce(outs[1], g_txt[d_idx], oouts[1])
                    li, d2 = inter_ce(outs[1], g_img, d_idx)
                    d_outs = [None, (d1 + d2) * interintra_weight]
                else:
                    B = img.shape[0]
                    l1, da = moon_ce(outs[0], g_img[d_idx], oouts[0], rows_norm=2 * B)     # intra: one CE over the 2B stacked rows
                    l2, db = moon_ce(outs[1], g_txt[d_idx], oouts[1], rows_norm=2 * B)
                    l3, da2 = inter_ce(outs[0], g_txt, d_idx)
                    l4, db2 = inter_ce(outs[1], g_img, d_idx)
                    d_outs = [(da + da2) * interintra_weight, (db + db2) * interintra_weight]
                grads = {k: g for k, g in O.backward(p, cfg, cache, d_outs).items() if k in tkeys}
                clip_grad_norm(grads, 2.0)
                adam_apply(p, grads, state, lr, weight_decay)
    return results


def pub_features(p, cfg, kind, pub_batches):
    """update_pub_feature: eval-mode feat_out features of the client's own modality over the public set, and the index list."""
    feats, index = [], []
    for img, ids, idx in pub_batches:
        outs, _ = _features(p, cfg, kind, img, ids)
        feats.append(outs[0] if kind == "img" else outs[1])
        index.extend(int(i) for i in idx)
    return torch.cat(feats), index


def aggregate_features(vecs: List[Tensor], G_other: Tensor) -> Optional[Tensor]:
    """creamflserver.py:373-405: per client log_prob diag of vec @ G_other.T, softmax over clients per sample, weighted sum."""
    if not vecs:
        return None
    w = []
    for v in vecs:
        logits = v @ G_other.t()
        log_prob = logits - torch.log(torch.exp(logits).sum(dim=1, keepdim=True))
        w.append(torch.diagonal(log_prob).reshape(1, -1))
    w = torch.softmax(torch.cat(w, dim=0), dim=0)
    return sum(v * w[i].reshape(-1, 1) for i, v in enumerate(vecs))


def zero_init_aggregate(global_keys: Sequence[str], uploads: Dict[int, Dict[str, Tensor]], ids: Sequence[int], coefficients):
    """creamflserver.py:257-288: final[k] = sum_i upload_i[k] * c_i[k] into zeros (no global term, no sequential blend)."""
    out = {}
    for k in global_keys:
        acc = None
        for i in ids:
            if k not in uploads[i] or coefficients[k][i] == 0:
                continue
            term = uploads[i][k].float() * coefficients[k][i]
            acc = term if acc is None else acc + term
        out[k] = acc
    return out


def kd_distill(p, cfg, pub_batches, distill_index, img_vec: Tensor, txt_vec: Tensor, kd_weight: float, p_lr: float):
    """creamflserver.py:294-336: per public batch loss = kd_weight * (MSE(out_img, img_vec[d_idx]) + MSE(out_txt, txt_vec[d_idx])),
    clip 2, AdamW(lr=p_lr) with torch's default weight_decay 0.01.  Mutates p."""
    dd = {int(b): a for a, b in enumerate(distill_index)}
    state = dict(m={}, v={}, t={})
    tkeys = set(O.trainable_keys(p, cfg))
    for img, ids, index in pub_batches:
        d_idx = torch.tensor([dd[int(i)] for i in index])
        outs, cache = O.forward(p, cfg, [img, ids])                 # rtv heads: unit-norm features either way
        d_outs = []
        for o, tgt in ((outs[0], img_vec[d_idx]), (outs[1], txt_vec[d_idx])):
            d_outs.append(kd_weight * 2.0 * (o - tgt) / o.numel())
        grads = {k: g for k, g in O.backward(p, cfg, cache, d_outs).items() if k in tkeys}
        clip_grad_norm(grads, 2.0)
        adam_apply(p, grads, state, p_lr, weight_decay=0.01)
