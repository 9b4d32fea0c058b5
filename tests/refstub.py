"""Import the *real* reference (/root/reference) in the build container, with stub modules for
its missing third-party deps (timm, torchvision, torchtext, torchmultimodal, wandb, ...).

Used only by tests/golden/make_golden.py (to generate committed golden vectors) and by
tests/test_oracle_vs_reference.py (skipped where /root/reference is absent, e.g. the GPU box).
Nothing from the reference is copied: it is imported in place, read-only.

Recipe follows SURVEY.md section 8(c).
"""
from __future__ import annotations

import os
import sys
import types
from unittest import mock

REF_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "models"))


_loaded = {}


def load_reference():
    """Returns a namespace with .mome, .fedavgserver, .fedavgclient, .create_model, .ContrastiveLoss."""
    if _loaded:
        return _loaded["ns"]
    if not reference_available():
        raise RuntimeError("reference not present")
    import torch
    import torch.nn as nn
    # (1) transformers first: its lazy loader chokes on a spec-less 'timm' stub
    import transformers.models.bert.modeling_bert  # noqa: F401

    registry = {}

    class DropPath(nn.Module):
        """timm 0.9.12 DropPath semantics (per-sample Bernoulli(keep) / keep, train only)."""

        def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
            super().__init__()
            self.drop_prob = drop_prob
            self.scale_by_keep = scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = x.new_empty(shape).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, std=1.0, **kw):
        return nn.init.trunc_normal_(t, std=std)

    def register_model(fn):
        registry[fn.__name__] = fn
        return fn

    def create_model(name, pretrained=False, **kw):
        return registry[name](pretrained, **kw)

    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    filler = mock.MagicMock()
    layer_names = ["PatchEmbed", "Mlp", "AttentionPoolLatent", "RmsNorm", "PatchDropout", "SwiGLUPacked",
                   "lecun_normal_", "resample_patch_embed", "resample_abs_pos_embed", "use_fused_attn",
                   "get_act_layer", "get_norm_layer", "LayerType"]
    timm = mk("timm", create_model=create_model)
    timm.layers = mk("timm.layers", DropPath=DropPath, trunc_normal_=trunc_normal_,
                     **{n: filler for n in layer_names})
    timm.models = mk("timm.models", create_model=create_model)
    timm.models.layers = mk("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
    timm.models.registry = mk("timm.models.registry", register_model=register_model)
    for name in ["wandb", "torchvision", "torchvision.datasets", "torchvision.transforms", "torchtext",
                 "torchtext.datasets", "pycocotools", "pycocotools.coco", "medmnist", "ml_collections", "ujson",
                 "torchmultimodal", "torchmultimodal.modules", "torchmultimodal.modules.losses",
                 "torchmultimodal.modules.losses.contrastive_loss_with_temperature"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import src.models.mome as mome
    import src.client.fedavgclient as fedavgclient
    import src.server.fedavgserver as fedavgserver
    import src.client.fedproxclient as fedproxclient
    import src.client.creamflclient as creamflclient
    import src.server.creamflserver as creamflserver

    ns = types.SimpleNamespace(mome=mome, fedavgclient=fedavgclient, fedavgserver=fedavgserver, fedproxclient=fedproxclient, creamflclient=creamflclient, creamflserver=creamflserver,
                               create_model=create_model, registry=registry, DropPath=DropPath)
    _loaded["ns"] = ns
    return ns


class RefArgs:
    """Minimal stand-in for the reference's argparse namespace (fields the hot path reads)."""

    def __init__(self, **kw):
        d = dict(vocab_size=7732, seq_len=32, dropout=0.0, shared_param="none", share_scope="dataset",
                 colearn_param="none", with_aux=False, aux_trained=False, aux_attn_only=False, aux_mlp_only=False,
                 optimizer="AdamW", lr=1e-4, weight_decay=0.0, momentum=0.0, nesterov=False, max_grad_norm=0.0,
                 E=1, B=4, no_shuffle=True, debug=False, distributed=False, mm_distributed=False,
                 compensation=False, out_modality_scales=[1, 1, 1], equal_sampled=True, C=0.25, K=8,
                 warmup_modality="none", warmup_rounds=0, freeze_modality="none", freeze_rounds=0,
                 lr_decay=0.99, lr_decay_step=1, num_thread=1, mp=False, seed=1, algorithm="fedavg",
                 datasets=[], modalities=[], pretrained=False, train_only=True, beta1=0.0,
                 eval_type="global", server_device="cpu", fedavg_eval=False, dataset="x")
        d.update(kw)
        self.__dict__.update(d)
