"""Host-side mirror of the reference's ``ModalityAgnosticTransformer`` (/root/reference/src/models/mome.py:671-922)
on top of the HIP library: same constructor arguments, ``forward([img|None, txt|None], feat_out)`` signature,
``state_dict()`` key names / shapes / order, ``required_params()`` / ``aux_params()`` and the factory registry
(``create_model('mome_small_patch16', ...)``, mome.py:924-1033).

All parameters live in ONE flat fp32 device buffer (layout from ``fc_model_segment``); the named parameters are views
into it, so the optimizer / aggregation collectives work on a single contiguous tensor.  Compute is entirely in
libfedcola_hip.so -- there is no PyTorch fallback: calling ``forward`` without a GPU raises.
"""
from __future__ import annotations

import copy
import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from ._lib import FcModelCfg, FcSegment, check, ptr

_TASKS = {None: _lib.FC_TASK_NONE, "cls": _lib.FC_TASK_CLS, "rtv": _lib.FC_TASK_RTV}


class _Handle:
    """Owns an fc_model_t*; exposes the segment table."""

    def __init__(self, cfg: FcModelCfg):
        self.cfg = cfg
        self.h = C.c_void_p()
        check(_lib.lib().fc_model_create(C.byref(cfg), C.byref(self.h)))
        L = _lib.lib()
        self.total = int(L.fc_model_num_params(self.h))
        self.segments: "OrderedDict[str, dict]" = OrderedDict()
        for i in range(L.fc_model_num_segments(self.h)):
            s = FcSegment()
            check(L.fc_model_segment(self.h, i, C.byref(s)))
            self.segments[s.name.decode()] = dict(index=i, offset=int(s.offset), numel=int(s.numel),
                                                  shape=tuple(int(s.shape[k]) for k in range(s.ndim)),
                                                  trainable=bool(s.trainable))

    def __del__(self):
        try:
            if self.h:
                _lib.lib().fc_model_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _MomeFn(torch.autograd.Function):
    """loss.backward() support: the flat parameter tensor is the only differentiable input."""

    @staticmethod
    def forward(ctx, flat, model, img, ids, feat_out, droppath):
        outs = model._run_forward(img, ids, feat_out, droppath)
        ctx.model = model
        ctx.keep = (img, ids, droppath)        # the C side keeps raw pointers to ids / droppath until fc_backward
        ctx.present = [o is not None for o in outs]
        return tuple(o if o is not None else flat.new_zeros(0) for o in outs)

    @staticmethod
    def backward(ctx, g_img, g_txt):
        model = ctx.model
        g = [g_img.contiguous().float() if ctx.present[0] else None, g_txt.contiguous().float() if ctx.present[1] else None]
        grads = torch.zeros_like(model.flat)
        L = _lib.lib()
        with torch.cuda.device(model.flat.device):
            check(L.fc_backward(model._handle.h, ptr(model.flat), ptr(model._wc_or_flat()), ptr(g[0]), ptr(g[1]), ptr(grads),
                                ptr(model._ws), model._ws.numel(), _lib.stream_ptr()))
        return grads, None, None, None, None, None


class ModalityAgnosticTransformer(nn.Module):
    def __init__(self, modalities, num_classes, tasks, shared_param="none", share_scope="dataset", colearn_param="none",
                 img_size=224, patch_size=16, in_chans=3, embed_dim=768, drop_rate=0.0, num_heads=12, vocab_size=30522,
                 max_text_len=40, mlp_ratio=4, qkv_bias=True, qk_scale=None, attn_drop_rate=0.0, drop_path_rate=0.0,
                 depth=12, shared_start_index=-1, layer_scale_init_values=None, precision="fp32", init=True, **kwargs):
        super().__init__()
        assert qkv_bias and qk_scale is None and attn_drop_rate == 0.0 and drop_rate == 0.0 and not layer_scale_init_values, \
            "only the configuration the reference factories use is implemented (qkv_bias, no attn/proj dropout, no LayerScale)"
        for m in modalities:
            if m not in ("img", "txt", None):
                raise NotImplementedError                                  # mome.py:720-721
        for t in tasks:
            if t not in ("cls", "rtv", None):
                raise NotImplementedError                                  # mome.py:762-763
        self.embed_dim = embed_dim
        self.with_aux = kwargs.get("with_aux", False)
        self.aux_trained = kwargs.get("aux_trained", False)
        self.aux_attn_only = kwargs.get("aux_attn_only", False)
        self.aux_mlp_only = kwargs.get("aux_mlp_only", False)
        if self.with_aux and None in modalities and self.aux_attn_only and self.aux_mlp_only:
            raise ValueError("Both aux_attn_only and aux_mlp_only cannot be True.")   # mome.py:779
        self.shared_start_index = depth if shared_start_index == -1 else shared_start_index
        self.shared_param, self.scope, self.colearn_param = shared_param, share_scope, colearn_param
        self.modalities = list(modalities)
        self.tasks = list(tasks)
        self.num_classes = list(num_classes)
        self.num_heads, self.depth = num_heads, depth
        self.drop_path_rate = float(drop_path_rate)
        self.dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]        # mome.py:726-728
        self.precision = precision
        self._hp = dict(img_size=img_size, patch_size=patch_size, in_chans=in_chans, vocab_size=vocab_size,
                        max_text_len=max_text_len, mlp_ratio=mlp_ratio)
        cfg = FcModelCfg(
            has_img=int(modalities[0] == "img"), has_txt=int(modalities[1] == "txt"), img_size=img_size, patch=patch_size,
            in_chans=in_chans, dim=embed_dim, depth=depth, heads=num_heads, mlp_hidden=int(embed_dim * mlp_ratio),
            vocab=vocab_size, max_text_len=max_text_len, task_img=_TASKS[tasks[0]], task_txt=_TASKS[tasks[1]],
            num_classes_img=int(num_classes[0] or 0), num_classes_txt=int(num_classes[1] or 0), with_aux=int(self.with_aux),
            aux_trained=int(self.aux_trained), aux_attn_only=int(self.aux_attn_only), aux_mlp_only=int(self.aux_mlp_only),
            precision=_lib.FC_PREC_BF16 if precision == "bf16" else _lib.FC_PREC_FP32,
            colearn_attn=int(colearn_param == "attn" and modalities[0] == "img" and modalities[1] == "txt"))
        self._handle = _Handle(cfg)
        self.flat = nn.Parameter(torch.zeros(self._handle.total, dtype=torch.float32))
        self._wc = None
        self._wc_version = -1
        self._ws = None
        self._views: Optional[Dict[str, nn.Parameter]] = None
        self._alias: Dict[str, str] = {}
        if init:
            self._reference_init()
        self.sync_shared_weights()                 # mome.py:815: the constructor ends with it

    # ------------------------------------------------------------------ parameter views
    @property
    def segments(self):
        return self._handle.segments

    def _view(self, name) -> torch.Tensor:
        s = self.segments[name]
        return self.flat.data[s["offset"]: s["offset"] + s["numel"]].view(s["shape"])

    @torch.no_grad()
    def _reference_init(self):
        """PyTorch default initialisation in the reference's construction order (mome.py:708-769), so that
        ``torch.manual_seed(s); Model(...)`` yields the same weights as the reference (pos_embed / cls_token zeros)."""
        hp, D = self._hp, self.embed_dim
        for i, mod in enumerate(self.modalities):
            if mod == "img":
                conv = nn.Conv2d(hp["in_chans"], D, kernel_size=hp["patch_size"], stride=hp["patch_size"])
                self._view(f"embeddings.{i}.embed.proj.weight").copy_(conv.weight)
                self._view(f"embeddings.{i}.embed.proj.bias").copy_(conv.bias)
            elif mod == "txt":
                pre = f"embeddings.{i}.text_embeddings"
                self._view(pre + ".word_embeddings.weight").copy_(nn.Embedding(hp["vocab_size"], D, padding_idx=0).weight)
                self._view(pre + ".position_embeddings.weight").copy_(nn.Embedding(hp["max_text_len"], D).weight)
                self._view(pre + ".token_type_embeddings.weight").copy_(nn.Embedding(2, D).weight)
                self._view(pre + ".LayerNorm.weight").fill_(1.0)
        Hd = int(D * hp["mlp_ratio"])
        for i, mod in enumerate(self.modalities):
            if mod is None:
                continue
            for l in range(self.depth):
                p = f"blockses.{i}.{l}"
                self._view(p + ".norm1.weight").fill_(1.0)
                self._view(p + ".norm2.weight").fill_(1.0)
                for nm, (o, n) in (("attn.qkv", (3 * D, D)), ("attn.proj", (D, D)), ("mlp.fc1", (Hd, D)), ("mlp.fc2", (D, Hd))):
                    lin = nn.Linear(n, o)                                   # always drawn: the reference builds every tower first
                    if f"{p}.{nm}.weight" not in self.segments:             # colearn 'attn': this module is dropped for the main tower's
                        continue
                    self._view(f"{p}.{nm}.weight").copy_(lin.weight)
                    self._view(f"{p}.{nm}.bias").copy_(lin.bias)
                    if f"{p}.{nm}.aux_weight" in self.segments:           # build_aux: aux_weight is the old layer's weight
                        self._view(f"{p}.{nm}.aux_weight").copy_(lin.weight)
        self._view("norm.weight").fill_(1.0)
        for i, t in enumerate(self.tasks):
            if t == "cls" and self.modalities[i] is not None and (self.num_classes[i] or 0) > 0:
                lin = nn.Linear(D, self.num_classes[i])
                self._view(f"heads.{i}.head.weight").copy_(lin.weight)
                self._view(f"heads.{i}.head.bias").copy_(lin.bias)
        self._bump()

    def _bump(self):
        self._wc_version = -1

    def sync_shared_weights(self):
        """mome.py:818-842.
        * scope == 'all': the reference puts the main tower's block list into the slot of every absent modality
          (``self.blockses[i] = self.blockses[main_idx]``), so ``state_dict()`` gains ``blockses.{i}.*`` keys that alias the main
          tower's tensors.  Here: alias keys that are views of the same flat segments (``_alias``).
        * colearn_param == 'blocks': the reference's loop only rebinds its loop variable (``blocks = self.blockses[main_idx]``,
          :832-835) -- no module changes hands.  Reproduced as the no-op it is.
        * colearn_param == 'attn': the other tower's Attention modules ARE replaced by the main tower's (:836-840): shared qkv / proj
          parameters, gradients from both towers.  The C layout gives the second tower no attention segments (fc_model_cfg.colearn_attn);
          here the second tower's keys alias the main tower's, as in the reference's state_dict."""
        main_idx = next(i for i, m in enumerate(self.modalities) if m is not None)
        self._alias = {}
        if self.scope == "all":
            for i, m in enumerate(self.modalities):
                if m is None:
                    self._alias[f"blockses.{i}."] = f"blockses.{main_idx}."
        if self.colearn_param == "attn":
            for i, m in enumerate(self.modalities):
                if m is not None and i != main_idx:
                    for l in range(self.depth):
                        self._alias[f"blockses.{i}.{l}.attn."] = f"blockses.{main_idx}.{l}.attn."

    def _alias_keys(self):
        """(alias key, target key) pairs created by sync_shared_weights (scope == 'all'), in the target's order."""
        out = []
        for apre, tpre in getattr(self, "_alias", {}).items():
            for k in self.segments:
                if k.startswith(tpre):
                    out.append((apre + k[len(tpre):], k))
        return out

    def pretrain_vit(self, model_strs, loader=None):
        """mome.py:788-816: map a timm ViT checkpoint per present modality onto this model's keys ('patch_embed' ->
        'embeddings.{i}.embed', 'blocks.' -> 'blockses.{i}.', cls_token / pos_embed -> 'embeddings.{i}.*'; for the 'ours' checkpoints
        'head' -> 'heads.head'), load non-strictly, then sync_shared_weights().  ``loader(model_str) -> state_dict`` stands in for
        ``timm.create_model(model_str, pretrained=True)`` / ``torch.load('pretrain.pt')``, which need files this build cannot fetch."""
        for i, model_str in enumerate(model_strs):
            if model_str is None:
                continue
            if loader is None:
                raise NotImplementedError("pretrained timm checkpoints are not available offline (mome.py:788-816): pass loader=")
            sd = dict(loader(model_str))
            if "ours" in model_str:
                for k, v in list(sd.items()):
                    if "head" in k:
                        sd[k.replace("head", "heads.head")] = v
            for k, v in list(sd.items()):
                if "patch_embed" in k:
                    sd[k.replace("patch_embed", f"embeddings.{i}.embed")] = v
                elif "blocks." in k:
                    sd[k.replace("blocks", f"blockses.{i}")] = v
            sd[f"embeddings.{i}.cls_token"] = sd["cls_token"]
            sd[f"embeddings.{i}.pos_embed"] = sd["pos_embed"]
            self.load_state_dict(sd, strict=False)
        self.sync_shared_weights()

    # ------------------------------------------------------------------ nn.Module surface
    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        if self._views is None:
            self._views = OrderedDict()
            for k, s in self.segments.items():
                p = nn.Parameter(self._view(k), requires_grad=s["trainable"])
                self._views[k] = p
        for k, p in self._views.items():
            g = self.flat.grad
            s = self.segments[k]
            # frozen segments keep grad None, like the reference's requires_grad=False parameters: torch optimizers skip only
            # parameters whose grad is None (a zero gradient would still be weight-decayed)
            if g is not None and s["trainable"]:
                p.grad = g[s["offset"]: s["offset"] + s["numel"]].view(s["shape"])
            elif not s["trainable"]:
                p.grad = None
            yield (prefix + ("." if prefix else "") + k, p)

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    _SLOTS = ("norm1.", "attn.qkv.", "attn.proj.", "norm2.", "mlp.fc1.", "mlp.fc2.")

    @classmethod
    def _canon(cls, k):
        """Sort key that puts alias keys where the reference's module traversal lists them (embeddings, blockses.{i}.{l} in
        norm1 / attn.qkv / attn.proj / norm2 / mlp.fc1 / mlp.fc2 order, norm, heads)."""
        parts = k.split(".")
        if parts[0] == "embeddings":
            return (0, int(parts[1]), 0, 0)
        if parts[0] == "blockses":
            rest = ".".join(parts[3:])
            slot = next((n for n, sl in enumerate(cls._SLOTS) if rest.startswith(sl)), len(cls._SLOTS))
            return (1, int(parts[1]), int(parts[2]), slot)
        return (2 if parts[0] == "norm" else 3, 0, 0, 0)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        items = [(k, k) for k in self.segments] + list(self._alias_keys())   # alias keys: views of the owner's segment
        if self._alias:
            items.sort(key=lambda kv: self._canon(kv[0]))                    # stable: keys of one slot keep their order
        for k, tk in items:
            sd[prefix + k] = self._view(tk)
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True, assign=False):
        alias = dict(self._alias_keys())
        missing = [k for k in self.segments if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self.segments and k not in alias and not k.endswith("position_ids")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, v in state_dict.items():
            k = alias.get(k, k)
            if k in self.segments:
                self._view(k).copy_(v.reshape(self.segments[k]["shape"]))
        self._bump()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def required_params(self):
        """mome.py:844-860 (views that alias the parameters)."""
        sd = self.state_dict()
        for i, mod in enumerate(self.modalities):
            if mod is None:
                for k in list(sd):
                    if f"blockses.{i}" in k:
                        sd.pop(k)
        if self.with_aux:
            for k in list(sd):
                if "aux" in k or "cross_modal_scale" in k:
                    sd.pop(k)
        return sd

    def aux_params(self):
        """mome.py:862-878."""
        if not self.with_aux:
            raise ValueError("No aux params.")
        none_idx = [i for i, m in enumerate(self.modalities) if m is None]
        return OrderedDict((k, v) for k, v in self.state_dict().items()
                           if "aux" in k and not any(f"blockses.{i}" in k for i in none_idx))

    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        self._views = None
        self._wc = None
        self._ws = None
        self._bump()
        return self

    def __deepcopy__(self, memo):
        new = ModalityAgnosticTransformer.__new__(ModalityAgnosticTransformer)
        nn.Module.__init__(new)
        for k, v in self.__dict__.items():
            if k in ("_parameters", "_buffers", "_modules", "_wc", "_ws", "_views", "_handle", "_dp_keep", "_agg_partial", "_agg_cat", "_client_owned"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._handle = _Handle(self._handle.cfg)
        for name, value in self.__dict__.get("_options", {}).items():    # carry over the handle's run-time switches
            if name != "gemm_form":
                check(_lib.lib().fc_model_set_option(new._handle.h, {"mlp_fused": _lib.FC_OPT_MLP_FUSED, "step_graph": _lib.FC_OPT_STEP_GRAPH}[name], value))
        for k, s in self.segments.items():                               # carry over freeze flags
            if not s["trainable"] and new._handle.segments[k]["trainable"]:
                new.set_trainable(k, False)
        new.flat = nn.Parameter(self.flat.data.clone())
        new._wc, new._ws, new._views, new._wc_version = None, None, None, -1
        new.train(self.training)
        return new

    def refresh_from(self, src) -> bool:
        """Make this model an exact copy of ``src`` WITHOUT building a new object: what ``copy.deepcopy(src)`` returns, when this model
        already has src's configuration (same library configuration struct, precision, device, segment list).  The library handle, the
        bf16 compute-weight buffer and the workspace stay allocated -- a FedavgClient's download() at every round (fedavgclient.py:155)
        otherwise pays a new handle, its device tables and a 3.6-GB workspace claim before its first step.  Returns False (nothing
        touched) when the two models are not interchangeable; the caller then deep-copies."""
        if type(src) is not type(self) or src is self:
            return False
        if self.flat.device != src.flat.device or self.flat.shape != src.flat.shape or self.flat.dtype != src.flat.dtype:
            return False
        if bytes(self._handle.cfg) != bytes(src._handle.cfg) or self.precision != src.precision:
            return False
        if list(self.segments.keys()) != list(src.segments.keys()):
            return False
        for k, s in src.segments.items():                                # freeze flags follow the source (as in __deepcopy__)
            if self.segments[k]["trainable"] != s["trainable"]:
                self.set_trainable(k, s["trainable"])
        skip = ("_parameters", "_buffers", "_modules", "_wc", "_ws", "_views", "_handle", "_dp_keep", "_agg_partial", "_agg_cat", "segments", "training",
                "_wc_version", "_client_owned")
        for k, v in src.__dict__.items():                                # plain Python state (hyper-parameters, alias maps, rates)
            if k not in skip:
                self.__dict__[k] = copy.deepcopy(v)
        with torch.no_grad():
            self.flat.data.copy_(src.flat.data)
        self.flat.grad = None
        self._views = None
        self._bump()                                                     # the compute weights are re-cast before the next forward
        self.train(src.training)
        return True

    def set_option(self, name: str, value: int):
        """Run-time switch of the library handle -- TOOLS BUILD ONLY (FC_PROBES_LIB=1; include/fedcola_hip.h, FC_OPT_*): "mlp_fused" (fc1 -> GELU ->
        fc2 as one launch per 64-row panel), "step_graph" (fc_client_step replays a captured HIP graph), "gemm_form" (process-wide tile form of
        under-filled launches: 0 | 64 | 3 | 4).  The round-5 experiments: exact, no faster in the ViT-S client step (profiles/r05), not in the product."""
        code = {"mlp_fused": _lib.FC_OPT_MLP_FUSED, "step_graph": _lib.FC_OPT_STEP_GRAPH, "gemm_form": _lib.FC_OPT_GEMM_FORM}[name]
        if not _lib.is_probes_build():
            raise _lib.FedcolaHipError(f"set_option({name!r}): the product library has no run-time options; the experiments live in the tools build "
                                       "(python -m fedcola_amd.build --probes, FC_PROBES_LIB=1)")
        check(_lib.lib().fc_model_set_option(self._handle.h, code, int(value)))
        opts = self.__dict__.setdefault("_options", {})
        opts[name] = int(value)
        if name == "mlp_fused":
            self._wc_version = -1                  # the packed weight streams are written by the next prepare_weights()

    def set_trainable(self, key: str, flag: bool):
        check(_lib.lib().fc_model_set_trainable(self._handle.h, self.segments[key]["index"], int(flag)))
        self.segments[key]["trainable"] = bool(flag)
        self._views = None

    # ------------------------------------------------------------------ device buffers
    def _require_gpu(self):
        if not self.flat.is_cuda:
            raise _lib.FedcolaHipError("ModalityAgnosticTransformer.forward needs the model on an MI355X (cuda) device: "
                                       "the hot path is HIP-only (no CPU fallback)")

    def _wc_or_flat(self):
        return self._wc if self._wc is not None else self.flat

    def prepare_weights(self, force=False):
        """(Re)build the compute weights (bf16 shadow / aux fold) if the parameters changed."""
        L = _lib.lib()
        nbytes = int(L.fc_compute_weights_bytes(self._handle.h))
        if nbytes == 0:
            return
        if self._wc is None or self._wc.device != self.flat.device or self._wc.numel() < nbytes:      # (grows when a tools-build option adds packed streams)
            self._wc = torch.empty(nbytes, dtype=torch.uint8, device=self.flat.device)
            force = True
        if force or self._wc_version != self.flat._version:
            check(L.fc_prepare_weights(self._handle.h, ptr(self.flat), ptr(self._wc), _lib.stream_ptr()))
            self._wc_version = self.flat._version

    def workspace(self, B: int, n_txt: int):
        need = int(_lib.lib().fc_workspace_bytes(self._handle.h, B, n_txt))
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.flat.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.flat.device)
        return self._ws

    def side_stream(self):
        """The library's auxiliary stream (text tower) as a torch stream: where the next batch's H2D copy belongs
        (fedcola_amd.loaders.DevicePrefetcher)."""
        h = _lib.lib().fc_model_side_stream(self._handle.h)
        if not h:
            raise _lib.FedcolaHipError(_lib.lib().fc_last_error().decode())
        return torch.cuda.ExternalStream(int(h), device=self.flat.device)

    def make_droppath(self, B: int, generator=None) -> Optional[torch.Tensor]:
        """timm DropPath multipliers [2, depth, 2, B] (0 or 1/keep), drawn on device; None when inactive (eval / rate 0)."""
        if not self.training or self.drop_path_rate <= 0.0:
            return None
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep.device != self.flat.device:      # constant table: uploaded once (a per-step H2D copy stalls the launch queue)
            keep = 1.0 - torch.tensor(self.dpr, dtype=torch.float32).view(1, self.depth, 1, 1).to(self.flat.device)
            self._dp_keep = keep
        u = torch.rand(2, self.depth, 2, B, device=self.flat.device, generator=generator)
        return ((u < keep).float() / keep).contiguous()

    # ------------------------------------------------------------------ forward
    def _run_forward(self, img, ids, feat_out, droppath):
        self._require_gpu()
        B = (img if img is not None else ids).shape[0]
        n_txt = ids.shape[1] if ids is not None else 0
        self.prepare_weights()
        ws = self.workspace(B, n_txt)
        dev = self.flat.device
        outs = [None, None]
        for i, mod in enumerate(self.modalities):
            if mod is None:
                continue
            width = self.embed_dim if (feat_out or self.tasks[i] == "rtv") else int(self.num_classes[i])
            outs[i] = torch.empty(B, width, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):      # the library's internal streams / tables belong to the current device
            check(_lib.lib().fc_forward(self._handle.h, ptr(self.flat), ptr(self._wc_or_flat()), ptr(img), ptr(ids), B, n_txt,
                                        int(bool(feat_out)), ptr(droppath), ptr(ws), ws.numel(), ptr(outs[0]), ptr(outs[1]),
                                        _lib.stream_ptr()))
        return outs

    def forward(self, x, feat_out=False, droppath=None):
        img, ids = x[0], x[1]
        for i, mod in enumerate(self.modalities):
            if mod is None:
                assert x[i] is None, "None modality should have None input."           # mome.py:890
        if img is not None:
            if img.dim() == 4 and img.shape[1] == 1:                                    # mome.py:893-894
                img = img.repeat(1, 3, 1, 1)
            H, W = img.shape[2], img.shape[3]
            assert H == self._hp["img_size"] and W == self._hp["img_size"], \
                f"Input image size ({H}*{W}) doesn't match model ({self._hp['img_size']}*{self._hp['img_size']})."   # mome.py:262
            img = img.contiguous().float()
        if ids is not None:
            ids = ids.contiguous().long()
        if droppath is None:
            B = (img if img is not None else ids).shape[0]
            droppath = self.make_droppath(B)
        if torch.is_grad_enabled() and self.flat.requires_grad:
            o = _MomeFn.apply(self.flat, self, img, ids, feat_out, droppath)
            return [o[i] if self.modalities[i] is not None else None for i in range(2)]
        return self._run_forward(img, ids, feat_out, droppath)


# ---------------------------------------------------------------------- factory registry (timm.create_model stand-in)
_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name, pretrained=False, **kwargs):
    """timm.create_model(model_str, pretrained=..., num_classes=[..], modalities=[..], args=, tasks=[..], with_aux=...)
    as called at fedavgserver.py:151-155."""
    if model_name not in _REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name})")
    return _REGISTRY[model_name](pretrained, **kwargs)


def _factory(embed_dim, depth, heads, timm_name=None):
    def build(pretrained, args, **kwargs):
        model = ModalityAgnosticTransformer(img_size=224, patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=heads,
                                            vocab_size=args.vocab_size, max_text_len=args.seq_len, drop_path_rate=args.dropout,
                                            shared_param=args.shared_param, share_scope=args.share_scope,
                                            colearn_param=args.colearn_param, precision=getattr(args, "precision", "fp32"),
                                            **kwargs)
        model.sync_shared_weights()
        if pretrained:     # mome.py:951-952: model.pretrain_vit([timm name, None]); the checkpoint comes from args.pretrain_loader here
            model.pretrain_vit([timm_name, None], loader=getattr(args, "pretrain_loader", None))
        return model
    return build


def _register(name, embed_dim, depth, heads, timm_name=None):
    fn = _factory(embed_dim, depth, heads, timm_name)
    fn.__name__ = fn.__qualname__ = name
    return register_model(fn)


mome_small_patch16 = _register("mome_small_patch16", 384, 12, 6, "vit_small_patch16_224")                              # mome.py:924-953
mome_tiny_patch16 = _register("mome_tiny_patch16", 192, 12, 3, "vit_tiny_patch16_224")                                 # mome.py:955-974
mome_small_patch16_224_in21k = _register("mome_small_patch16_224_in21k", 384, 12, 6, "vit_small_patch16_224_in21k")    # mome.py:976-995
mome_toy_patch16_224 = _register("mome_toy_patch16_224", 4, 1, 2)                                                      # mome.py:1016-1033
# The reference's only 768-wide factory (mome_base_patch16_224_ours, mome.py:998-1014) is broken (passes share_strategy=,
# skips sync_shared_weights); this build defines the D=768, H=12 model by analogy.
mome_base_patch16 = _register("mome_base_patch16", 768, 12, 12, "vit_base_patch16_224")
