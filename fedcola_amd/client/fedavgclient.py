"""``FedavgClient`` with the reference's surface (/root/reference/src/client/fedavgclient.py:15-190): same constructor,
``download / update / upload / evaluate / __len__ / __repr__``, same ``update()`` result schema
``{epoch: {'loss': sum(loss_b*|b|)/len(training_set), 'metrics': {...}}}``.

What changed underneath: the batch-loop body (zero_grad, forward, criterion, backward, AdamW.step) is ONE call into
libfedcola_hip (``fc_client_step``); weights, gradients and optimizer state live in flat device buffers for the whole round
(no per-round H2D/D2H of the model, no per-step ``loss.item()`` sync: the loss sum is accumulated on the device and read
once per epoch)."""
from __future__ import annotations

import copy
import inspect
import logging

import torch

from .. import _lib
from .._lib import check, ptr
from ..criterions import CRITERIA
from ..utils import MetricManager
from .baseclient import BaseClient

logger = logging.getLogger(__name__)


class FedavgClient(BaseClient):
    def __init__(self, args, training_set, test_set, task="cls", eval_metrics=["acc1"], modality="ct", writer=None,
                 criterion="CrossEntropyLoss"):
        super().__init__()
        self.args = args
        self.training_set = training_set
        self.test_set = test_set
        self.optim = torch.optim.__dict__[self.args.optimizer]
        self.criterion = CRITERIA[criterion] if criterion in CRITERIA else torch.nn.__dict__[criterion]
        self.train_loader = self._create_dataloader(self.training_set, shuffle=not self.args.no_shuffle, test=False)
        self.test_loader = self._create_dataloader(self.test_set, shuffle=False, test=True)
        self.task = task
        self.modality = modality
        self.eval_metrics = eval_metrics
        self.writer = writer
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.dataset = None

    def _refine_optim_args(self, args):
        """fedavgclient.py:34-42: optimizer kwargs = the optimizer's named args that ``args`` defines."""
        required_args = inspect.getfullargspec(self.optim)[0]
        return {a: getattr(args, a) for a in required_args if hasattr(args, a)}

    def _create_dataloader(self, dataset, shuffle, test=True):
        if self.args.B == 0:
            self.args.B = len(self.training_set)
        # device clients: same sampling as the DataLoader below, threaded assembly into pinned buffers (loaders/batch.py); the
        # reference's single-process DataLoader (fedavgclient.py:44-53) needs 80-160 ms per B = 64 image batch on the MI355X host
        # against a 4.5-ms device step.  Default for datasets that are a pure function of the index: those that offer get_batch()
        # (in-memory / pre-decoded) and the caption datasets (Flickr30kCap / CocoCaptionsCap, also behind Subset wrappers) whose
        # transform is deterministic -- the reference's --resize/--imnorm chain is -- which are decoded ONCE per client into a
        # loaders.cache.DecodedCache (uint8 when lossless) and served from memory.  Random-transform datasets keep the DataLoader
        # (worker threads would reorder their RNG draws).  args.fast_loader = True / False and args.decode_cache = False override.
        from ..loaders.cache import DecodedCache
        cuda = torch.cuda.is_available()
        # (nothing is fetched -- no RNG draw -- unless the fast loader and the cache are both wanted: args.fast_loader = False or
        # args.decode_cache = False must leave the reference's sample and shuffle stream untouched)
        want_fast = getattr(self.args, "fast_loader", None)
        cacheable = want_fast is not False and getattr(self.args, "decode_cache", cuda) and not hasattr(dataset, "get_batch") and DecodedCache.applicable(dataset)
        if getattr(self.args, "fast_loader", cuda and (hasattr(dataset, "get_batch") or cacheable)):
            from ..loaders.batch import PinnedBatchLoader
            if cacheable:
                dataset = DecodedCache(dataset, workers=getattr(self.args, "loader_workers", 8))      # built at its first batch
            # the training loop reads its batches through DevicePrefetcher, which expands a cache's uint8 image codes on the device
            # (loader.device_finish): the batch crosses PCIe as 9.6 MB instead of 38.5 MB and the host does no arithmetic
            raw = cacheable and not test and cuda and getattr(self.args, "prefetch", True)
            return PinnedBatchLoader(dataset, self.args.B, shuffle=shuffle, workers=getattr(self.args, "loader_workers", 8), raw=raw)
        return torch.utils.data.DataLoader(dataset=dataset, batch_size=self.args.B, shuffle=shuffle)

    # ------------------------------------------------------------------ the hot loop (fedavgclient.py:55-116)
    # ---- hooks for subclasses that interleave other optimizer steps with the fused ones (creamflclient.py)
    def _before_update(self, st):
        pass

    def _after_epoch(self, e, st, step):
        return step

    def _fused_step_ok(self, st):
        return True

    def _count_step(self, st, skipped):
        pass

    def _segmented_step(self, st, *a):
        raise NotImplementedError

    def _prox(self):
        """(global flat copy, mu) for a proximal term, or None -- overridden by FedproxClient."""
        return None

    def update(self):
        mm = MetricManager(self.eval_metrics) if self.modality != "img+txt" else MetricManager([])
        model = self.model
        model.train()
        model.to(self.device)
        prox = self._prox()
        oargs = self._refine_optim_args(self.args)
        fused = self.args.optimizer in ("AdamW", "SGD") and not getattr(self.args, "distributed", False) and \
            not getattr(self.args, "mm_distributed", False) and not getattr(self.args, "force_unfused", False)
        sgd = self.args.optimizer == "SGD"       # main.py:269: the argument's default; composed on the device like the clipped AdamW step
        if not fused:
            return self._update_unfused(mm, oargs, prox)
        dev = model.flat.device
        # fedavgclient.py:98-99: clip_grad_norm_ sits between backward and optimizer.step, so the step is composed from the ABI's pieces
        # (fc_forward / criterion / fc_backward / fc_prox_term / fc_clip_grad_norm / fc_adamw_step) instead of the one fused call
        max_norm = float(getattr(self.args, "max_grad_norm", 0) or 0)
        # the first epoch's loader is started NOW: its first batch is assembled (and any sampler RNG is drawn -- nothing below draws from
        # the CPU generator) while the optimizer state is allocated; fedavgclient.py:79 creates the iterator at the loop head
        started = None
        if dev.type == "cuda" and getattr(self.args, "prefetch", True) and self.args.E > 0:
            from ..loaders.prefetch import DevicePrefetcher       # H2D of the next batches on a copy stream (N4)
            started = DevicePrefetcher(self.train_loader, dev, depth=2, stream=model.side_stream(), device_ring=self._device_ring()).start()
        n = model.flat.numel()
        grads = torch.zeros(n, device=dev)
        exp_avg = torch.zeros(n, device=dev)        # optimizer re-created every round: fresh state (fedavgclient.py:63)
        exp_avg_sq = torch.zeros(n, device=dev)
        lossbuf = torch.zeros(2, device=dev)
        lr = float(oargs.get("lr", 1e-3))
        if sgd:
            momentum, nesterov = float(oargs.get("momentum", 0.0)), bool(oargs.get("nesterov", False))
            if nesterov and (momentum <= 0 or float(oargs.get("dampening", 0.0)) != 0):
                raise ValueError("Nesterov momentum requires a momentum and zero dampening")      # torch.optim.SGD's own check
            if float(oargs.get("dampening", 0.0)) != 0 or oargs.get("maximize", False):
                return self._update_unfused(mm, oargs, prox)
        betas = oargs.get("betas", (0.9, 0.999))
        eps = float(oargs.get("eps", 1e-8))
        wd = float(oargs.get("weight_decay", 0.0 if sgd else 1e-2))      # torch's defaults: SGD 0, AdamW 1e-2
        L = _lib.lib()
        if prox is not None:
            prox_scratch = torch.empty(L.fc_prox_scratch_bytes(model._handle.h), dtype=torch.uint8, device=dev)
        clip_scratch = torch.empty(L.fc_clip_scratch_bytes(model._handle.h), dtype=torch.uint8, device=dev) if max_norm > 0 else None
        step = 0
        # optimizer state of this round, shared with subclasses' extra steps (CreamFL's public-set distillation)
        st = dict(grads=grads, exp_avg=exp_avg, exp_avg_sq=exp_avg_sq, lr=lr, betas=betas, eps=eps, wd=wd, dev=dev, steps_done=0)
        self._before_update(st)
        for e in range(self.args.E):
            num = 0
            lossbuf.zero_()
            broke = False
            loader = self.train_loader
            if started is not None:
                loader, started = started, None
            elif dev.type == "cuda" and getattr(self.args, "prefetch", True):
                from ..loaders.prefetch import DevicePrefetcher
                loader = DevicePrefetcher(self.train_loader, dev, depth=2, stream=model.side_stream(), device_ring=self._device_ring())
            raw_finish = getattr(self.train_loader, "device_finish", None)
            for batch in loader:
                if raw_finish is not None and torch.is_tensor(batch[0]) and batch[0].dtype == torch.uint8:
                    # a raw (uint8 image code) loader without the prefetcher in front of it (args.prefetch switched off after construction,
                    # a non-cuda device): expand here -- the codes must never reach `.float()` below as if they were pixels
                    batch = list(batch)
                    batch[0] = batch[0].to(dev, non_blocking=True)
                    batch = raw_finish(batch)
                if num >= 2 and self.args.debug:                       # fedavgclient.py:73-75
                    mm.add_loss_sum(lossbuf[0].clone())
                    mm.aggregate(num * self.args.B, e + 1)
                    broke = True
                    break
                if self.modality == "img+txt":
                    inputs, targets = batch[0], batch[1]
                    img = inputs.to(dev, non_blocking=True).contiguous().float()
                    ids = targets.to(dev, non_blocking=True).contiguous().long()
                    labels = None
                elif self.modality == "img":
                    img = batch[0].to(dev, non_blocking=True).contiguous().float()
                    if img.dim() == 4 and img.shape[1] == 1:
                        img = img.repeat(1, 3, 1, 1)
                    ids, labels = None, batch[1].to(dev, non_blocking=True).contiguous().long()
                else:
                    ids = batch[0].to(dev, non_blocking=True).contiguous().long()
                    img, labels = None, batch[1].to(dev, non_blocking=True).contiguous().long()
                B = (img if img is not None else ids).shape[0]
                n_txt = ids.shape[1] if ids is not None else 0
                model.prepare_weights()
                ws = model.workspace(B, n_txt)
                dp = model.make_droppath(B)
                step += 1
                if not self._fused_step_ok(st):
                    self._segmented_step(st, img, ids, labels, B, n_txt, dp, ws, lossbuf)
                    model._wc_version = model.flat._version
                    self._collect_metrics(mm, ws, B, labels)
                    num += 1
                    continue
                if max_norm > 0 or sgd:
                    self._forward_backward(st, img, ids, labels, B, n_txt, dp, ws, lossbuf)
                    if prox is not None:
                        check(L.fc_prox_term(model._handle.h, ptr(model.flat), ptr(prox[0]), float(prox[1]), B, ptr(grads), ptr(lossbuf),
                                             ptr(prox_scratch), prox_scratch.numel(), _lib.stream_ptr()))
                    if max_norm > 0:
                        check(L.fc_clip_grad_norm(model._handle.h, ptr(grads), max_norm, ptr(clip_scratch), clip_scratch.numel(), None,
                                                  _lib.stream_ptr()))
                    if sgd:                                            # exp_avg doubles as the momentum buffer (fresh every round, like the optimizer)
                        check(L.fc_sgd_step(model._handle.h, ptr(model.flat), ptr(grads), ptr(exp_avg), lr, momentum, int(nesterov), wd, step,
                                            _lib.stream_ptr()))
                    else:
                        check(L.fc_adamw_step(model._handle.h, ptr(model.flat), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), lr, float(betas[0]),
                                              float(betas[1]), eps, wd, step, _lib.stream_ptr()))
                    model._bump()                                      # the compute weights are rebuilt by the next prepare_weights()
                    st["steps_done"] = step
                    self._count_step(st, None)
                    self._collect_metrics(mm, ws, B, labels)
                    num += 1
                    continue
                step_args = (model._handle.h, ptr(model.flat), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), ptr(model._wc_or_flat()),
                             ptr(img), ptr(ids), ptr(labels), B, n_txt, ptr(dp), lr, float(betas[0]), float(betas[1]), eps, wd,
                             step, ptr(lossbuf), ptr(ws), ws.numel(), _lib.stream_ptr())
                if prox is None:
                    check(L.fc_client_step(*step_args))
                else:                                                   # fedproxclient.py:64-67 inside the same fused step
                    check(L.fc_client_step_prox(*step_args, ptr(prox[0]), float(prox[1]), ptr(prox_scratch), prox_scratch.numel()))
                model._wc_version = model.flat._version          # fc_client_step refreshed the compute weights itself
                st["steps_done"] = step
                self._count_step(st, None)
                self._collect_metrics(mm, ws, B, labels)
                num += 1
            if not broke:
                mm.add_loss_sum(lossbuf[0].clone())                    # sum_b loss_b*|b| accumulated on the device
                mm.aggregate(len(self.training_set), e + 1)
                res = mm.results[e + 1]
                logger.info(f'[Client {self.id}] loss: {res["loss"]}')
            step = self._after_epoch(e, st, step)
        # the reference moves the model back to the CPU here (fedavgclient.py:114); weights stay resident in HBM instead
        return mm.results

    def _collect_metrics(self, mm, ws, B, labels):
        """acc1 etc. for uni-modal clients: the logits of the step's forward are copied out of the workspace."""
        if not mm.metric_funcs:
            return
        model = self.model
        i = 0 if self.modality == "img" else 1
        logits = torch.empty(B, model.num_classes[i], device=model.flat.device)
        check(_lib.lib().fc_copy_outputs(model._handle.h, ptr(ws), ws.numel(), ptr(logits) if i == 0 else None,
                                         ptr(logits) if i == 1 else None, _lib.stream_ptr()))
        for module in mm.metric_funcs.values():
            module.collect(logits, labels)

    def _forward_backward(self, st, img, ids, labels, B, n_txt, dp, ws, lossbuf):
        """optimizer.zero_grad(); forward; criterion; loss.backward() (fedavgclient.py:79-97) from the ABI's separate entry points:
        the gradients are left in st['grads'] for whatever has to happen before the optimizer step."""
        model, L = self.model, _lib.lib()
        dev = st["dev"]
        st["grads"].zero_()
        i = 0 if self.modality == "img" else 1
        if self.modality == "img+txt":
            D = model.embed_dim
            oi, ot = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
            check(L.fc_forward(model._handle.h, ptr(model.flat), ptr(model._wc_or_flat()), ptr(img), ptr(ids), B, n_txt, 1, ptr(dp), ptr(ws),
                               ws.numel(), ptr(oi), ptr(ot), _lib.stream_ptr()))
            da, db = torch.empty_like(oi), torch.empty_like(ot)
            scratch = torch.empty(L.fc_contrastive_scratch_floats(B), device=dev)
            from ..criterions import contrastive_tau
            check(L.fc_contrastive_loss_fwd_bwd(ptr(oi), ptr(ot), B, D, contrastive_tau(), ptr(scratch), scratch.numel(), ptr(lossbuf), ptr(da),
                                                ptr(db), _lib.stream_ptr()))
            d0, d1 = da, db
        else:
            C_ = model.num_classes[i]
            lg = torch.empty(B, C_, device=dev)
            check(L.fc_forward(model._handle.h, ptr(model.flat), ptr(model._wc_or_flat()), ptr(img), ptr(ids), B, n_txt, 0, ptr(dp), ptr(ws),
                               ws.numel(), ptr(lg) if i == 0 else None, ptr(lg) if i == 1 else None, _lib.stream_ptr()))
            dl = torch.empty_like(lg)
            check(L.fc_ce_loss_fwd_bwd(ptr(lg), ptr(labels), B, C_, ptr(lossbuf), ptr(dl), _lib.stream_ptr()))
            d0, d1 = (dl, None) if i == 0 else (None, dl)
        check(L.fc_backward(model._handle.h, ptr(model.flat), ptr(model._wc_or_flat()), ptr(d0), ptr(d1), ptr(st["grads"]), ptr(ws), ws.numel(),
                            _lib.stream_ptr()))

    def _device_ring(self):
        """Fixed device buffers for the prefetcher (three per position: the same addresses step after step), which is what lets a
        step_graph model replay its captured graphs.  Only then: otherwise the prefetcher allocates per batch from torch's caching
        allocator and a client holds no HBM between rounds."""
        if not getattr(self.model, "_options", {}).get("step_graph"):
            return None
        ring = getattr(self, "_dev_ring", None)
        if ring is None:
            ring = self._dev_ring = {}
        return ring

    def _update_unfused(self, mm, oargs, prox=None):
        """Any torch optimizer / gradient clipping: HIP forward+backward through autograd, torch.optim on the flat views."""
        model = self.model
        dev = model.flat.device
        optimizer = self.optim(model.parameters(), **oargs)
        finish = getattr(self.train_loader, "device_finish", None)      # a raw loader's uint8 image codes -> floats (loaders/cache.py)
        for e in range(self.args.E):
            num = 0
            broke = False
            for batch in self.train_loader:
                if num >= 2 and self.args.debug:
                    mm.aggregate(num * self.args.B, e + 1)
                    broke = True
                    break
                if finish is not None and self.modality != "txt":
                    batch = finish([batch[0].to(dev)] + list(batch[1:]))
                model.flat.grad = None
                if self.modality == "img+txt":
                    inputs, targets = batch[0].to(dev), batch[1].to(dev)
                    outputs = model([inputs, targets], feat_out=True)
                    loss = self.criterion()(*outputs)
                elif self.modality == "img":
                    inputs, targets = batch[0].to(dev), batch[1].to(dev)
                    outputs = model([inputs, None])[0]
                    loss = self.criterion()(outputs, targets)
                else:
                    inputs, targets = batch[0].to(dev), batch[1].to(dev)
                    outputs = model([None, inputs])[1]
                    loss = self.criterion()(outputs, targets)
                if prox is not None:                                    # fedproxclient.py:64-67 (un-squared per-tensor norms)
                    term = 0.
                    for k, sgm in model.segments.items():
                        if not sgm["trainable"]:
                            continue
                        sl = slice(sgm["offset"], sgm["offset"] + sgm["numel"])
                        term = term + (model.flat[sl] - prox[0][sl]).norm(2)
                    loss = loss + prox[1] * (0.5 * term)
                loss.backward()
                params = list(model.parameters())                      # refreshes the .grad views
                if getattr(self.args, "max_grad_norm", 0) > 0:
                    torch.nn.utils.clip_grad_norm_([model.flat], self.args.max_grad_norm)
                optimizer.step()
                model._bump()
                if self.modality != "img+txt":
                    mm.track(loss.item(), outputs.detach(), targets)
                else:
                    mm.track(loss.item(), outputs[0].detach())
                num += 1
            if not broke:
                mm.aggregate(len(self.training_set), e + 1)
        return mm.results

    @torch.inference_mode()
    def evaluate(self):
        """fedavgclient.py:118-153 ("Not used" in the reference)."""
        if self.args.train_only:
            return {"loss": -1, "metrics": {"none": -1}}
        mm = MetricManager(self.eval_metrics)
        self.model.eval()
        self.model.to(self.device)
        for inputs, targets in self.test_loader:
            inputs, targets = inputs.to(self.device), targets.to(self.device)
            x = [inputs, None] if self.modality == "img" else [None, inputs]
            outputs = self.model(x)[0 if self.modality == "img" else 1]
            loss = self.criterion()(outputs, targets)
            mm.track(loss.item(), outputs, targets)
        mm.aggregate(len(self.test_set))
        return mm.results

    # ---- client models are recycled.  The reference builds a new deep copy of the global model at every download() and the server drops
    # it after the round (``client.model = None``, fedavgserver.py:501,673).  Here a dropped device model is parked in a small per-process
    # pool and download() refreshes a parked (or the current) model of the same configuration in place: same values as the deep copy,
    # without a new library handle, compute-weight buffer and workspace per client and round (0.8 ms + a slow first step of a 100-ms round).
    # Consequence: tensors handed out by upload() / state_dict() of a client model are views that stay valid until a later download()
    # recycles that model (the server consumes them within the round); ``args.recycle_models = False`` restores a fresh deep copy per round.
    _POOL = []
    _POOL_MAX = 8

    @property
    def model(self):
        return self._BaseClient__model

    @model.setter
    def model(self, model):
        old = self.__dict__.get("_BaseClient__model")
        # only a model this class built in download() may be recycled: an object somebody assigned (a global model, say) is never written to
        if model is None and old is not None and getattr(old, "_client_owned", False) and getattr(old, "flat", None) is not None and old.flat.is_cuda:
            if len(FedavgClient._POOL) < FedavgClient._POOL_MAX and not any(m is old for m in FedavgClient._POOL):
                FedavgClient._POOL.append(old)
        self._BaseClient__model = model

    def download(self, models):
        """fedavgclient.py:155-156 (a device-to-device clone of the flat buffer; into a recycled model object when one fits)."""
        src = models[self.dataset]
        if getattr(self.args, "recycle_models", True) and hasattr(src, "refresh_from"):
            cur = self.__dict__.get("_BaseClient__model")
            if cur is not None and getattr(cur, "_client_owned", False) and cur.refresh_from(src):
                return
            for i, m in enumerate(FedavgClient._POOL):
                if m.refresh_from(src):
                    del FedavgClient._POOL[i]
                    m._client_owned = True
                    self._BaseClient__model = m
                    return
        m = copy.deepcopy(src)
        if hasattr(m, "refresh_from"):
            m._client_owned = True
        self._BaseClient__model = m

    def upload(self):
        """fedavgclient.py:158-184.  Returns the state_dict (device tensors); with ``with_aux`` on a uni-modal client every
        re-param linear's weight is folded (W + A*s, ``fc_upload_fold``) and the aux / scale keys are dropped."""
        model = self.model
        sd = model.state_dict()
        if self.args.with_aux and self.modality != "img+txt":
            if self.args.aux_attn_only and self.args.aux_mlp_only:
                raise ValueError("Both aux_attn_only and aux_mlp_only cannot be True.")
            folded = torch.empty_like(model.flat.data)
            check(_lib.lib().fc_upload_fold(model._handle.h, ptr(model.flat), ptr(folded), _lib.stream_ptr()))
            self._folded = folded                                       # keep alive: the views below alias it
            new_sd = {}
            for k, s in model.segments.items():
                if "aux" in k or "cross_modal_scale" in k:
                    continue
                new_sd[k] = folded[s["offset"]: s["offset"] + s["numel"]].view(s["shape"])
            return new_sd
        return sd

    def __len__(self):
        return len(self.training_set)

    def __repr__(self):
        return f"CLIENT < {self.id} >"
