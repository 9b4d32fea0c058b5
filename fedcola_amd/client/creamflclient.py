"""``CreamflClient`` (/root/reference/src/client/creamflclient.py:19-247): FedAvg local epochs, each followed by one contrastive
distillation pass over the server's public set, plus ``update_pub_feature`` (the client's features of the public set).

Per public batch (creamflclient.py:146-233), all on the device:
  features of the current model (train mode) and of the round-start model (``old_model``, eval)     fc_forward(feat_out)
  loss = (moon|intra + inter) * interintra_weight, temperature 0.5                                     fc_cream_moon_loss, fc_cream_inter_loss
  backward of the feature gradients                                                                   fc_backward
  clip_grad_norm_(parameters, 2)                                                                      fc_clip_grad_norm
  optimizer.step() -- the SAME AdamW as the local epochs (moments and per-parameter step counts)      fc_adamw_step_segs
torch's optimizer skips parameters whose gradient is None and keeps a step count per parameter: a classification head gets no
gradient from the feature losses, so its step count lags after the first distillation pass; the fused ``fc_client_step``
(one global step) is used only while all counts agree, otherwise the step is composed from the same entry points."""
from __future__ import annotations

import copy
import logging

import numpy as np
import torch
from torch.utils import data

from .. import _lib
from .._lib import check, ptr
from .fedavgclient import FedavgClient

logger = logging.getLogger(__name__)


class CreamflClient(FedavgClient):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def get_pub_loader(self, dataset, batch_size=512):
        """creamflclient.py:23-36"""
        return data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False, drop_last=False)

    # ------------------------------------------------------------------ features of the public set
    def _features(self, model, images, captions, train, B):
        """feat_out features through the C ABI: returns (img [B, D] | None, txt [B, D] | None) device tensors."""
        dev = model.flat.device
        L = _lib.lib()
        D = model.embed_dim
        use_img, use_txt = self.modality != "txt", self.modality != "img"
        img = images.to(dev).contiguous().float() if use_img else None
        if img is not None and img.dim() == 4 and img.shape[1] == 1:
            img = img.repeat(1, 3, 1, 1)
        ids = captions.to(dev).contiguous().long() if use_txt else None
        n_txt = ids.shape[1] if ids is not None else 0
        model.prepare_weights()
        ws = model.workspace(B, n_txt)
        dp = model.make_droppath(B) if train else None
        oi = torch.empty(B, D, device=dev) if use_img else None
        ot = torch.empty(B, D, device=dev) if use_txt else None
        check(L.fc_forward(model._handle.h, ptr(model.flat), ptr(model._wc_or_flat()), ptr(img), ptr(ids), B, n_txt, 1, ptr(dp), ptr(ws), ws.numel(),
                           ptr(oi), ptr(ot), _lib.stream_ptr()))
        return oi, ot, ws, (img, ids, dp)

    @torch.no_grad()
    def update_pub_feature(self):
        """creamflclient.py:38-66: eval-mode features of the client's own modality over the public set."""
        self.model.to(self.device)
        self.model.eval()
        feature, distill_index = [], []
        for images, captions, _, _, index in self.get_pub_loader(self.pub_dataset, batch_size=self.args.pub_batch_size):
            B = images.shape[0]
            oi, ot, _, _ = self._features(self.model, images, captions, False, B)
            feature.append((oi if self.modality == "img" else ot).clone())
            distill_index.extend(index)
        self.pub_features = torch.cat(feature, dim=0)
        self.distill_index = distill_index

    # ------------------------------------------------------------------ hooks of FedavgClient.update
    def _before_update(self, st):
        if self.args.optimizer != "AdamW":
            raise NotImplementedError("CreamflClient is implemented on the fused HIP path (AdamW, no max_grad_norm on the local epochs)")
        model = self.model
        self.old_model = copy.deepcopy(model)                 # creamflclient.py:74 (device-to-device clone of the flat buffer)
        self.old_model.eval()
        segs = list(model.segments.items())
        st["seg_t"] = np.zeros(len(segs), dtype=np.int32)      # torch's per-parameter step counts
        st["seg_train"] = np.array([1 if s["trainable"] else 0 for _, s in segs], dtype=np.int32)
        # parameters the feature losses never reach: the classification heads (feat_out returns before them, mome.py:915)
        st["seg_feat"] = np.array([0 if k.startswith("heads.") else 1 for k, _ in segs], dtype=np.int32) * st["seg_train"]

    def _fused_step_ok(self, st):
        t = st["seg_t"][st["seg_train"] == 1]
        return t.size == 0 or bool((t == t[0]).all())

    def _count_step(self, st, mask):
        st["seg_t"] += st["seg_train"] if mask is None else mask

    def _adam_segs(self, st, mask):
        model = self.model
        self._count_step(st, mask)
        steps = np.ascontiguousarray(st["seg_t"] * mask, dtype=np.int32)
        check(_lib.lib().fc_adamw_step_segs(model._handle.h, ptr(model.flat), ptr(st["grads"]), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]),
                                            st["lr"], float(st["betas"][0]), float(st["betas"][1]), st["eps"], st["wd"],
                                            steps.ctypes.data, len(steps), ptr(model._wc) if model._wc is not None else None,
                                            _lib.stream_ptr()))
        model._wc_version = model.flat._version

    def _segmented_step(self, st, img, ids, labels, B, n_txt, dp, ws, lossbuf):
        """A local step composed from forward / criterion / backward / per-segment AdamW (step counts differ between parameters)."""
        self._forward_backward(st, img, ids, labels, B, n_txt, dp, ws, lossbuf)
        self._adam_segs(st, st["seg_train"])

    def _after_epoch(self, e, st, step):
        """The public-set distillation pass of creamflclient.py:133-233."""
        args, model = self.args, self.model
        if not (args.interintra_weight > 0 and not (args.no_mm_contrastive and self.modality == "img+txt")):
            return step
        L = _lib.lib()
        dev = st["dev"]
        w = float(args.interintra_weight)
        distill_dict = {int(b): a for a, b in enumerate(self.distill_index)}
        g_img = self.global_img_feature.to(dev).float().contiguous()
        g_txt = self.global_txt_feature.to(dev).float().contiguous()
        P, D = g_img.shape
        model.train()
        clip_scratch = torch.empty(L.fc_clip_scratch_bytes(model._handle.h), dtype=torch.uint8, device=dev)
        lossbuf = torch.zeros(2, device=dev)                         # distillation losses are not part of update()'s result
        sp = _lib.stream_ptr()
        for images, captions, _, _, index in self.get_pub_loader(self.pub_dataset, batch_size=args.pub_batch_size):
            B = images.shape[0]
            d_idx = torch.tensor([distill_dict[int(i)] for i in index.tolist()], dtype=torch.int64, device=dev)
            with torch.no_grad():
                ooi, oot, _, _ = self._features(self.old_model, images, captions, False, B)
            oi, ot, ws, keep = self._features(model, images, captions, True, B)      # last: fc_backward pairs with this forward
            scratch = torch.empty(L.fc_cream_inter_scratch_floats(B, P), device=dev)
            tgt = torch.empty(B, D, device=dev)
            d_img = torch.empty(B, D, device=dev) if oi is not None else None
            d_txt = torch.empty(B, D, device=dev) if ot is not None else None
            rows = 2 * B if self.modality == "img+txt" else B       # the img+txt client stacks both modalities into one CE (:207-217)
            for f, old, same, other, df in ((oi, ooi, g_img, g_txt, d_img), (ot, oot, g_txt, g_img, d_txt)):
                if f is None:
                    continue
                check(L.fc_gather_rows(ptr(same), ptr(d_idx), B, D, ptr(tgt), sp))
                check(L.fc_cream_moon_loss(ptr(f), ptr(tgt), ptr(old), B, D, rows, w, ptr(lossbuf), ptr(df), 0, sp))
                check(L.fc_cream_inter_loss(ptr(f), ptr(other), ptr(d_idx), B, P, D, w, ptr(scratch), scratch.numel(), ptr(lossbuf), ptr(df), 1, sp))
            st["grads"].zero_()                                      # optimizer.zero_grad()
            check(L.fc_backward(model._handle.h, ptr(model.flat), ptr(model._wc_or_flat()), ptr(d_img), ptr(d_txt), ptr(st["grads"]), ptr(ws),
                                ws.numel(), sp))
            check(L.fc_clip_grad_norm(model._handle.h, ptr(st["grads"]), 2.0, ptr(clip_scratch), clip_scratch.numel(), None, sp))
            self._adam_segs(st, st["seg_feat"])
            step += 1
            torch.cuda.current_stream().synchronize()                # temporaries above are released per batch
        return step

    def _update_unfused(self, mm, oargs, prox=None):
        raise NotImplementedError("CreamflClient is implemented on the fused HIP path (AdamW, no max_grad_norm on the local epochs)")

    def update(self):
        res = super().update()
        self.old_model = None
        return res
