"""CPU oracle for the FedCola client-step hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``fedcola_amd``) never does.

This is a restatement, in explicit forward *and* explicit backward formulas
(no autograd), of the reference's per-client training step:

  * ``ModalityAgnosticTransformer.forward``          /root/reference/src/models/mome.py:881-922
  * ``ImageEmbedding`` / ``PatchEmbed``               mome.py:597-611, 260-266
  * ``TextEmbedding`` (HF ``BertEmbeddings``)          mome.py:613-639
  * ``Block`` / ``Attention`` / ``Mlp``                mome.py:225-228, 150-168, 117-123
  * ``CrossModalReparamLinear``                        mome.py:58-60
  * ``ClassificationHead`` / ``RetrievalHead``         mome.py:647-649, 657-659
  * contrastive loss (torchmultimodal ``ContrastiveLossWithTemperature``;
    call site src/client/fedavgclient.py:95, registration src/criterions/__init__.py:3,8)
  * ``nn.CrossEntropyLoss``                            fedavgclient.py:85,90
  * ``torch.optim.AdamW`` step                         fedavgclient.py:63,100

Pinning status: every function here is checked against the *imported reference
itself* (stub recipe in tests/refstub.py) by tests/test_oracle_vs_reference.py in
the build container, and against the committed golden vectors in tests/golden/
(generated from the reference by tests/golden/make_golden.py) everywhere else.
Exception: the contrastive loss lives in the un-vendored, un-pinned third-party
``torchmultimodal`` (absent from the reference tree and from requirments.txt) —
**parity unpinned** for that one function; its published algorithm is restated
and pinned by closed-form known-answer tests (tests/test_oracle_kat.py).

All math is plain torch on CPU in the dtype of the inputs (fp32 by default,
fp64 for tight self-checks).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class OracleCfg:
    modalities: Tuple[Optional[str], Optional[str]] = ("img", "txt")
    tasks: Tuple[Optional[str], Optional[str]] = ("rtv", "rtv")
    num_classes: Tuple[Optional[int], Optional[int]] = (None, None)
    img_size: int = 224
    patch: int = 16
    in_chans: int = 3
    D: int = 384
    depth: int = 12
    heads: int = 6
    mlp_ratio: int = 4
    vocab: int = 30522
    max_text_len: int = 40
    with_aux: bool = False
    aux_trained: bool = False
    aux_attn_only: bool = False
    aux_mlp_only: bool = False
    colearn_attn: bool = False      # colearn_param == 'attn' (mome.py:836-840): every other tower's Attention module IS the main tower's

    @property
    def n_patches(self) -> int:
        return (self.img_size // self.patch) ** 2

    @property
    def aux_layers(self) -> Tuple[str, ...]:
        # mome.py:771-786 (build_aux); only when a modality is None (mome.py:768)
        if not (self.with_aux and None in self.modalities):
            return ()
        if self.aux_attn_only:
            return ("attn.qkv", "attn.proj")
        if self.aux_mlp_only:
            return ("mlp.fc1", "mlp.fc2")
        return ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")


# --------------------------------------------------------------------------- bf16 emulation of the throughput mode
# The product's bf16 mode (precision='bf16', the mode bench.py times) keeps activations and compute weights in bf16 and
# accumulates in fp32.  `with emulate_bf16():` makes this oracle round at exactly the points where the HIP kernels store
# or pack bf16 (DESIGN.md section 4 lists them), so that the timed path can be held to a tight tolerance instead of the
# loose "8 mantissa bits, 12 layers deep" bound.  Outside the context R() is the identity and nothing changes.
_EMULATE = [False]


class emulate_bf16:
    def __enter__(self):
        self.prev = _EMULATE[0]
        _EMULATE[0] = True

    def __exit__(self, *a):
        _EMULATE[0] = self.prev


def R(t: Tensor) -> Tensor:
    """Round to bf16 (nearest even) and back, inside `emulate_bf16()`; identity otherwise."""
    return t.to(torch.bfloat16).to(t.dtype) if _EMULATE[0] else t


LN_EPS_BLOCK = 1e-5   # nn.LayerNorm default, mome.py:199,203,215
LN_EPS_FINAL = 1e-6   # mome.py:751
LN_EPS_BERT = 1e-12   # BertConfig.layer_norm_eps default (mome.py:618-626)


# --------------------------------------------------------------------------- primitives
def ln_fwd(x: Tensor, g: Tensor, b: Tensor, eps: float):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * g + b, (xhat, rstd)


def ln_bwd(dy: Tensor, g: Tensor, saved):
    xhat, rstd = saved
    dg = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxhat = dy * g
    dx = rstd * (dxhat - dxhat.mean(-1, keepdim=True) - xhat * (dxhat * xhat).mean(-1, keepdim=True))
    return dx, dg, db


def gelu_fwd(x: Tensor) -> Tensor:
    # nn.GELU() default approximate='none' (erf form), mome.py:106,113
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def gelu_grad(x: Tensor) -> Tensor:
    return 0.5 * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0)))) + x * torch.exp(-0.5 * x * x) * (1.0 / math.sqrt(2.0 * math.pi))


def linear_fwd(x: Tensor, W: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ W.t()
    return y if b is None else y + b


def linear_bwd(dy: Tensor, x: Tensor, W: Tensor):
    dy2 = dy.reshape(-1, dy.shape[-1])
    x2 = x.reshape(-1, x.shape[-1])
    return dy @ W, dy2.t() @ x2, dy2.sum(0)


def patchify(img: Tensor, patch: int) -> Tensor:
    """[B,C,H,W] -> [B, (H/p)(W/p), C*p*p] in Conv2d weight order (c, ph, pw); patch order row-major
    (== conv output .flatten(2).transpose(1,2), mome.py:598-599)."""
    B, C, H, W = img.shape
    gh, gw = H // patch, W // patch
    x = img.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


# --------------------------------------------------------------------------- model forward
def _lin_weight(p: Dict[str, Tensor], prefix: str):
    """Effective weight of a (possibly re-parameterised) linear: W + s*A  (mome.py:58-60)."""
    W = p[prefix + ".weight"]
    if prefix + ".aux_weight" in p:
        return W + p[prefix + ".cross_modal_scale"] * p[prefix + ".aux_weight"]
    return W


def block_fwd(p, pre: str, x: Tensor, heads: int, dp1: Optional[Tensor], dp2: Optional[Tensor], apre: Optional[str] = None):
    """Block.forward mome.py:225-228.  dp1/dp2: per-sample drop-path multipliers [B] (already /keep).
    R(): bf16 storage points of the throughput mode (identity unless emulate_bf16()).
    apre: block whose Attention module this block uses (colearn_param == 'attn': the main tower's), default its own."""
    apre = apre or pre
    B, N, D = x.shape
    d = D // heads
    scale = d ** -0.5
    h1, s1 = ln_fwd(x, p[pre + ".norm1.weight"], p[pre + ".norm1.bias"], LN_EPS_BLOCK)
    h1 = R(h1)
    Wqkv = R(_lin_weight(p, apre + ".attn.qkv"))
    qkv = R(linear_fwd(h1, Wqkv, p[apre + ".attn.qkv.bias"]))
    qkv5 = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)          # mome.py:153
    q, k, v = qkv5[0] * scale, qkv5[1], qkv5[2]                            # mome.py:156
    S = q @ k.transpose(-2, -1)                                           # mome.py:157 (fp32)
    if _EMULATE[0]:   # the kernel packs exp(S - max) to bf16 for the PV product and divides by the fp32 row sum afterwards
        mx = S.max(-1, keepdim=True).values
        e = torch.exp(S - mx)
        l = e.sum(-1, keepdim=True)
        P = e / l
        O4 = (R(e) @ v) / l
    else:
        P = torch.softmax(S, dim=-1)                                      # mome.py:162
        O4 = P @ v
    O = R(O4.transpose(1, 2).reshape(B, N, D))                            # mome.py:165
    Wproj = R(_lin_weight(p, apre + ".attn.proj"))
    a = linear_fwd(O, Wproj, p[apre + ".attn.proj.bias"])
    if dp1 is not None:
        a = a * dp1.view(B, 1, 1)
    x1 = R(x + a)
    h2, s2 = ln_fwd(x1, p[pre + ".norm2.weight"], p[pre + ".norm2.bias"], LN_EPS_BLOCK)
    h2 = R(h2)
    W1 = R(_lin_weight(p, pre + ".mlp.fc1"))
    u = linear_fwd(h2, W1, p[pre + ".mlp.fc1.bias"])
    gact = R(gelu_fwd(u))
    gp = R(gelu_grad(u)) if _EMULATE[0] else None                         # the bf16 mode saves gelu'(u), not u
    W2 = R(_lin_weight(p, pre + ".mlp.fc2"))
    m = linear_fwd(gact, W2, p[pre + ".mlp.fc2.bias"])
    if dp2 is not None:
        m = m * dp2.view(B, 1, 1)
    x2 = R(x1 + m)
    cache = dict(s1=s1, h1=h1, q=q, k=k, v=v, P=P, O=O, s2=s2, h2=h2, u=u, gact=gact, gp=gp,
                 dp1=dp1, dp2=dp2, Wqkv=Wqkv, Wproj=Wproj, W1=W1, W2=W2)
    return x2, cache


def _lin_grads(grads, p, prefix: str, dW_eff: Tensor, db: Tensor, aux_trained: bool):
    """Route d(W_eff) to weight / aux_weight / cross_modal_scale  (autograd of mome.py:58-60)."""
    _acc(grads, prefix + ".weight", dW_eff)
    _acc(grads, prefix + ".bias", db)
    if prefix + ".aux_weight" in p:
        A = p[prefix + ".aux_weight"]
        s = p[prefix + ".cross_modal_scale"]
        _acc(grads, prefix + ".cross_modal_scale", (dW_eff * A).sum().reshape(1))
        if aux_trained:
            _acc(grads, prefix + ".aux_weight", s * dW_eff)


def _acc(grads, k, g):
    grads[k] = g if k not in grads else grads[k] + g


def block_bwd(p, pre: str, dx2: Tensor, c, heads: int, grads, aux_trained: bool, apre: Optional[str] = None, tap: Optional[dict] = None):
    """tap (optional, out): the block's intermediate dY tensors (dm, du, dx1, da, dqkv) for layer-local parity tests."""
    apre = apre or pre            # shared Attention module: its gradients accumulate under the owner's keys
    B, N, D = dx2.shape
    d = D // heads
    scale = d ** -0.5
    # ---- MLP branch
    dm = dx2 if c["dp2"] is None else R(dx2 * c["dp2"].view(B, 1, 1))
    dg, dW2, db2 = linear_bwd(dm, c["gact"], c["W2"])
    _lin_grads(grads, p, pre + ".mlp.fc2", dW2, db2, aux_trained)
    du = R(dg * (c["gp"] if c["gp"] is not None else gelu_grad(c["u"])))
    dh2, dW1, db1 = linear_bwd(du, c["h2"], c["W1"])
    dh2 = R(dh2)
    _lin_grads(grads, p, pre + ".mlp.fc1", dW1, db1, aux_trained)
    dx1n, dg2, dbb2 = ln_bwd(dh2, p[pre + ".norm2.weight"], c["s2"])
    _acc(grads, pre + ".norm2.weight", dg2)
    _acc(grads, pre + ".norm2.bias", dbb2)
    dx1 = R(dx2 + dx1n)
    # ---- attention branch
    da = dx1 if c["dp1"] is None else R(dx1 * c["dp1"].view(B, 1, 1))
    dO, dWp, dbp = linear_bwd(da, c["O"], c["Wproj"])
    dO = R(dO)
    _lin_grads(grads, p, apre + ".attn.proj", dWp, dbp, aux_trained)
    dO4 = dO.reshape(B, N, heads, d).transpose(1, 2)                      # [B,H,N,d]
    P, q, k, v = c["P"], c["q"], c["k"], c["v"]
    dP = dO4 @ v.transpose(-2, -1)
    if _EMULATE[0]:   # delta from the stored (rounded) output rows; P and dS are packed to bf16 as MFMA operands
        O4 = c["O"].reshape(B, N, heads, d).transpose(1, 2)
        delta = (dO4 * O4).sum(-1, keepdim=True)
        dS = R(P * (dP - delta))
        dV = R(P).transpose(-2, -1) @ dO4
    else:
        dS = P * (dP - (dP * P).sum(-1, keepdim=True))
        dV = P.transpose(-2, -1) @ dO4
    dq = (dS @ k) * scale                                                 # q was pre-scaled
    dk = dS.transpose(-2, -1) @ q
    dqkv = R(torch.stack([dq, dk, dV], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * D))
    if tap is not None:
        tap.update(dm=dm, du=du, dx1=dx1, da=da, dqkv=dqkv)
    dh1, dWq, dbq = linear_bwd(dqkv, c["h1"], c["Wqkv"])
    dh1 = R(dh1)
    _lin_grads(grads, p, apre + ".attn.qkv", dWq, dbq, aux_trained)
    dxn, dg1, dbb1 = ln_bwd(dh1, p[pre + ".norm1.weight"], c["s1"])
    _acc(grads, pre + ".norm1.weight", dg1)
    _acc(grads, pre + ".norm1.bias", dbb1)
    return R(dx1 + dxn)


def _attn_owner(cfg: OracleCfg, i: int, l: int) -> Optional[str]:
    if not cfg.colearn_attn:
        return None
    main = next(j for j, m in enumerate(cfg.modalities) if m is not None)     # mome.py:819-822
    return f"blockses.{main}.{l}"


def resolve_colearn(p: Dict[str, Tensor], cfg: OracleCfg) -> Dict[str, Tensor]:
    """state_dict-keyed weights -> named_parameters-keyed weights of a colearn_param == 'attn' model.  The reference's state_dict
    lists the shared Attention tensors under BOTH towers' keys; load_state_dict copies key by key, so the value under the LATER
    key (the other tower's) is what the shared tensor holds afterwards.  named_parameters() de-duplicates to the owner's key."""
    if not cfg.colearn_attn:
        return p
    main = next(j for j, m in enumerate(cfg.modalities) if m is not None)
    out = dict(p)
    for k in list(p):
        for i, m in enumerate(cfg.modalities):
            pre = f"blockses.{i}."
            if m is not None and i != main and k.startswith(pre) and ".attn." in k:
                out[f"blockses.{main}." + k[len(pre):]] = out.pop(k)
    return out


def forward(p: Dict[str, Tensor], cfg: OracleCfg, x: Sequence[Optional[Tensor]], feat_out: bool = False,
            dp_masks: Optional[Dict[Tuple[int, int, int], Tensor]] = None):
    """ModalityAgnosticTransformer.forward (mome.py:881-922).

    x = [img|None, txt|None]; returns (outs, cache).  dp_masks[(tower, layer, branch)] -> [B]
    multipliers (0 or 1/keep_prob) for timm DropPath; None => identity (eval or drop rate 0)."""
    outs: List[Optional[Tensor]] = [None, None]
    cache: Dict = {"towers": {}}
    for i, mod in enumerate(cfg.modalities):
        if mod is None:
            assert x[i] is None, "None modality should have None input."   # mome.py:890
            continue
        tc: Dict = {}
        if mod == "img":
            img = x[i]
            if img.dim() == 4 and img.shape[1] == 1:                      # mome.py:893-894
                img = img.repeat(1, 3, 1, 1)
            assert img.shape[2] == cfg.img_size and img.shape[3] == cfg.img_size   # mome.py:262
            pt = R(patchify(img, cfg.patch))
            Wp = R(p[f"embeddings.{i}.embed.proj.weight"].reshape(cfg.D, -1))
            tok = linear_fwd(pt, Wp, p[f"embeddings.{i}.embed.proj.bias"])
            B = tok.shape[0]
            cls = p[f"embeddings.{i}.cls_token"].expand(B, -1, -1)
            h = R(torch.cat([cls, tok], 1) + p[f"embeddings.{i}.pos_embed"])
            tc["patches"] = pt
            tc["Wp"] = Wp
        else:
            ids = x[i]
            pre = f"embeddings.{i}.text_embeddings"
            Nt = ids.shape[1]
            e = p[pre + ".word_embeddings.weight"][ids] + p[pre + ".token_type_embeddings.weight"][0] \
                + p[pre + ".position_embeddings.weight"][:Nt]
            h, tc["emb_ln"] = ln_fwd(e, p[pre + ".LayerNorm.weight"], p[pre + ".LayerNorm.bias"], LN_EPS_BERT)
            h = R(h)
            tc["ids"] = ids
        tc["blocks"] = []
        for l in range(cfg.depth):
            dp1 = dp_masks.get((i, l, 0)) if dp_masks else None
            dp2 = dp_masks.get((i, l, 1)) if dp_masks else None
            h, bc = block_fwd(p, f"blockses.{i}.{l}", h, cfg.heads, dp1, dp2, _attn_owner(cfg, i, l))
            tc["blocks"].append(bc)
        feats, tc["final_ln"] = ln_fwd(h, p["norm.weight"], p["norm.bias"], LN_EPS_FINAL)   # mome.py:906
        f = feats[:, 0]
        tc["shape"] = h.shape
        if feat_out or cfg.tasks[i] == "rtv":                             # mome.py:915 / 657-659
            nrm = f.norm(dim=-1, keepdim=True)
            outs[i] = f / nrm
            tc["head"] = ("rtv", nrm, outs[i])
        elif cfg.tasks[i] == "cls":                                       # mome.py:647-649
            outs[i] = linear_fwd(f, p[f"heads.{i}.head.weight"], p[f"heads.{i}.head.bias"])
            tc["head"] = ("cls", f)
        cache["towers"][i] = tc
    return outs, cache


def backward(p: Dict[str, Tensor], cfg: OracleCfg, cache, d_outs: Sequence[Optional[Tensor]]) -> Dict[str, Tensor]:
    """Explicit backward of ``forward``; returns grads keyed like state_dict (only keys that receive grad)."""
    grads: Dict[str, Tensor] = {}
    for i, mod in enumerate(cfg.modalities):
        if mod is None or d_outs[i] is None:
            continue
        tc = cache["towers"][i]
        B, N, D = tc["shape"]
        dout = d_outs[i]
        if tc["head"][0] == "rtv":
            _, nrm, o = tc["head"]
            df = (dout - o * (o * dout).sum(-1, keepdim=True)) / nrm
        else:
            _, f = tc["head"]
            df, dWh, dbh = linear_bwd(dout, f, p[f"heads.{i}.head.weight"])
            _acc(grads, f"heads.{i}.head.weight", dWh)
            _acc(grads, f"heads.{i}.head.bias", dbh)
        dfeats = torch.zeros(B, N, D, dtype=dout.dtype)
        dfeats[:, 0] = df
        dh, dgn, dbn = ln_bwd(dfeats, p["norm.weight"], tc["final_ln"])
        dh = R(dh)
        _acc(grads, "norm.weight", dgn)                                   # shared by both towers
        _acc(grads, "norm.bias", dbn)
        for l in reversed(range(cfg.depth)):
            dh = block_bwd(p, f"blockses.{i}.{l}", dh, tc["blocks"][l], cfg.heads, grads, cfg.aux_trained, _attn_owner(cfg, i, l))
        if mod == "img":
            _acc(grads, f"embeddings.{i}.pos_embed", dh.sum(0, keepdim=True))
            _acc(grads, f"embeddings.{i}.cls_token", dh[:, 0].sum(0).reshape(1, 1, D))
            dtok = dh[:, 1:]
            _, dWp, dbp = linear_bwd(dtok, tc["patches"], tc["Wp"])
            _acc(grads, f"embeddings.{i}.embed.proj.weight", dWp.reshape(p[f"embeddings.{i}.embed.proj.weight"].shape))
            _acc(grads, f"embeddings.{i}.embed.proj.bias", dbp)
        else:
            pre = f"embeddings.{i}.text_embeddings"
            de, dg, db = ln_bwd(dh, p[pre + ".LayerNorm.weight"], tc["emb_ln"])
            _acc(grads, pre + ".LayerNorm.weight", dg)
            _acc(grads, pre + ".LayerNorm.bias", db)
            ids = tc["ids"]
            Nt = ids.shape[1]
            dword = torch.zeros_like(p[pre + ".word_embeddings.weight"])
            dword.index_add_(0, ids.reshape(-1), de.reshape(-1, D))
            dword[0] = 0                                                  # padding_idx=0 row gets no grad
            _acc(grads, pre + ".word_embeddings.weight", dword)
            dpos = torch.zeros_like(p[pre + ".position_embeddings.weight"])
            dpos[:Nt] = de.sum(0)
            _acc(grads, pre + ".position_embeddings.weight", dpos)
            dtype_ = torch.zeros_like(p[pre + ".token_type_embeddings.weight"])
            dtype_[0] = de.reshape(-1, D).sum(0)
            _acc(grads, pre + ".token_type_embeddings.weight", dtype_)
    return grads


# --------------------------------------------------------------------------- losses
def contrastive_tau(dtype=torch.float32) -> float:
    """exp(clamp(log(1/0.07), log 1, log 100)) evaluated in fp32 like the upstream nn.Parameter."""
    ls = torch.tensor(math.log(1 / 0.07), dtype=torch.float32).clamp(math.log(1.0), math.log(100.0))
    return float(torch.exp(ls))


def contrastive_loss(a: Tensor, b: Tensor, tau: Optional[float] = None):
    """torchmultimodal ContrastiveLossWithTemperature, single process (no gather):
    L = 0.5*(CE(tau*a@b.T, arange) + CE(tau*b@a.T, arange)).  Returns (loss, da, db).  PARITY UNPINNED
    (third-party, absent from the reference tree) -- see module docstring."""
    tau = contrastive_tau() if tau is None else tau
    Bn = a.shape[0]
    L = tau * (a @ b.t())
    lse_r = torch.logsumexp(L, dim=1)
    lse_c = torch.logsumexp(L, dim=0)
    diag = L.diagonal()
    loss = 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())
    Pr = torch.exp(L - lse_r[:, None])
    Pc = torch.exp(L - lse_c[None, :])
    dL = (0.5 / Bn) * (Pr + Pc - 2.0 * torch.eye(Bn, dtype=a.dtype))
    return loss, tau * (dL @ b), tau * (dL.t() @ a)


def cross_entropy(logits: Tensor, y: Tensor):
    """nn.CrossEntropyLoss() (mean reduction).  Returns (loss, dlogits)."""
    Bn = logits.shape[0]
    lse = torch.logsumexp(logits, dim=1)
    loss = (lse - logits[torch.arange(Bn), y]).mean()
    dl = torch.exp(logits - lse[:, None])
    dl[torch.arange(Bn), y] -= 1.0
    return loss, dl / Bn


# --------------------------------------------------------------------------- optimizer
def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0):
    """One torch.optim.AdamW step (amsgrad=False), in place on p, m, v.  ``step`` is 1-based."""
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def sgd_step(p: Tensor, g: Tensor, buf, lr: float, momentum: float = 0.0, nesterov: bool = False, weight_decay: float = 0.0,
             dampening: float = 0.0):
    """One torch.optim.SGD step, in place on p; returns the momentum buffer (None in: the parameter's first step -> buf = clone(g)).
    The reference builds it through _refine_optim_args (fedavgclient.py:34-42, 63): lr, momentum, nesterov, weight_decay from args."""
    if weight_decay != 0.0:
        g = g.add(p, alpha=weight_decay)
    if momentum != 0.0:
        if buf is None:
            buf = g.clone()
        else:
            buf.mul_(momentum).add_(g, alpha=1.0 - dampening)
        g = g.add(buf, alpha=momentum) if nesterov else buf
    p.add_(g, alpha=-lr)
    return buf


# --------------------------------------------------------------------------- whole client step
def trainable_keys(p: Dict[str, Tensor], cfg: OracleCfg) -> List[str]:
    """Keys of nn.Parameters with requires_grad (aux_weight only when aux_trained; buffers excluded)."""
    out = []
    for k in p:
        if k.endswith("position_ids"):
            continue
        if k.endswith("aux_weight") and not cfg.aux_trained:
            continue
        out.append(k)
    return out


def prox_term(p: Dict[str, Tensor], g: Dict[str, Tensor], mu: float, keys: Sequence[str]):
    """FedproxClient.update's proximal term (src/client/fedproxclient.py:64-67):
        prox = sum over named parameters of ||param - global_param||_2   (per-tensor, UN-squared);   loss += mu * (0.5 * prox)
    Returns (mu*0.5*prox, {key: d/dparam}).  torch's norm backward yields 0 where the norm is 0 (first step: param == global)."""
    total = 0.0
    grads = {}
    for k in keys:
        d = p[k] - g[k]
        nrm = d.norm(2)
        total = total + nrm
        grads[k] = (mu * 0.5) * d / nrm if float(nrm) > 0 else torch.zeros_like(d)
    return mu * (0.5 * total), grads


def client_step(p, cfg: OracleCfg, batch, state, lr: float, weight_decay: float = 0.0, dp_masks=None, prox=None, sgd=None):
    """One iteration of FedavgClient.update's batch loop (fedavgclient.py:79-102):
    zero_grad -> fwd -> loss -> bwd -> AdamW.step (sgd = (momentum, nesterov): torch.optim.SGD.step instead, buffers in state['m']).
    ``state`` = {'step': int, 'm': {k}, 'v': {k}}.
    batch: ('img+txt', img, ids) | ('img', img, y) | ('txt', ids, y).  Returns (loss, outs, grads).
    prox = (global_params, mu): FedproxClient.update (fedproxclient.py:64-67) -- the proximal term joins loss and grads."""
    kind = batch[0]
    if kind == "img+txt":
        outs, cache = forward(p, cfg, [batch[1], batch[2]], feat_out=True, dp_masks=dp_masks)
        loss, da, db = contrastive_loss(outs[0], outs[1])
        d_outs = [da, db]
    elif kind == "img":
        outs, cache = forward(p, cfg, [batch[1], None], dp_masks=dp_masks)
        loss, dl = cross_entropy(outs[0], batch[2])
        d_outs = [dl, None]
    else:
        outs, cache = forward(p, cfg, [None, batch[1]], dp_masks=dp_masks)
        loss, dl = cross_entropy(outs[1], batch[2])
        d_outs = [None, dl]
    grads = backward(p, cfg, cache, d_outs)
    if prox is not None:
        keys = [k for k in p if p[k].dtype.is_floating_point]          # named_parameters(): every nn.Parameter
        pv, pg = prox_term(p, prox[0], prox[1], keys)
        loss = loss + pv
        for k, v in pg.items():
            if k in grads:
                grads[k] = grads[k] + v
            elif float(v.abs().max()) > 0:
                grads[k] = v
    state["step"] += 1
    for k in trainable_keys(p, cfg):
        if k not in grads:          # parameter without grad: torch optimizers skip it
            continue
        if sgd is not None:
            state["m"][k] = sgd_step(p[k], grads[k], state["m"].get(k), lr, momentum=sgd[0], nesterov=sgd[1], weight_decay=weight_decay)
            continue
        if k not in state["m"]:
            state["m"][k] = torch.zeros_like(p[k])
            state["v"][k] = torch.zeros_like(p[k])
        adamw_step(p[k], grads[k], state["m"][k], state["v"][k], state["step"], lr, weight_decay=weight_decay)
    return loss, outs, grads


def client_step_autograd(p, cfg: OracleCfg, batch, state, lr: float, weight_decay: float = 0.0):
    """The same img+txt step with the gradients taken by torch.autograd over forward() instead of the explicit backward() -- what the
    reference does on the CPU (loss.backward(), fedavgclient.py:97).  Used by bench.py's cpu_baseline (the faster of the two forms is
    reported) and checked against client_step in tests/test_oracle_kat.py."""
    assert batch[0] == "img+txt"
    keys = [k for k in trainable_keys(p, cfg)]
    leaves = {k: p[k].detach().requires_grad_(True) for k in keys}
    q = dict(p)
    q.update(leaves)
    outs, _ = forward(q, cfg, [batch[1], batch[2]], feat_out=True)
    tau = contrastive_tau(outs[0].dtype)
    la = (outs[0] @ outs[1].t()) * tau
    lab = torch.arange(la.shape[0])
    loss = (torch.nn.functional.cross_entropy(la, lab) + torch.nn.functional.cross_entropy(la.t(), lab)) / 2
    gl = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: g for k, g in zip(keys, gl) if g is not None}
    for k in grads:
        if k.endswith("word_embeddings.weight"):        # nn.Embedding(padding_idx=0): the padding row receives no gradient
            grads[k] = grads[k].clone()
            grads[k][0] = 0
    state["step"] += 1
    with torch.no_grad():
        for k in keys:
            if k not in grads:
                continue
            if k not in state["m"]:
                state["m"][k] = torch.zeros_like(p[k])
                state["v"][k] = torch.zeros_like(p[k])
            adamw_step(p[k], grads[k], state["m"][k], state["v"][k], state["step"], lr, weight_decay=weight_decay)
    return loss.detach(), [o.detach() for o in outs], grads


def drop_path_rates(rate: float, depth: int) -> List[float]:
    """dpr = linspace(0, rate, depth) (mome.py:726-728)."""
    return [x.item() for x in torch.linspace(0, rate, depth)]
