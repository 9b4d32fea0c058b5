"""``FedavgServer`` with the reference's surface (/root/reference/src/server/fedavgserver.py:117-898) for the hot path:
model construction through the factory registry, the per-key scope table, client sampling with Python ``random``
(bit-exact with the reference), the request fan-out, ``_aggregate`` (device blend + RCCL all-reduce, fedcola_amd/aggregate.py),
the aux-weight refresh, LR decay, and ``finalize``'s checkpoint.

Scaling model: one process per GPU (``torch.distributed`` over RCCL).  Every rank holds all global models and the full client
list (bookkeeping is replicated and deterministic); the sampled clients are dealt to ranks by their position in the sorted
sample, exactly like the reference deals them to ``cuda:(i % n_gpu)`` (fedavgserver.py:310-311).  Out of scope here (the
reference's control plane): wandb logging.  Central evaluation follows fedavgserver.py:676-760 with the retrieval evaluator of
``fedcola_amd.metrics.eval_coco`` (HIP ranking).
"""
from __future__ import annotations

import gc
import json
import logging
import os
import random
from collections import defaultdict
from importlib import import_module

import torch

from .. import aggregate as agg
from ..mome import create_model
from .baseserver import BaseServer

logger = logging.getLogger(__name__)

DATASET_2_TASK = {"BraTS": "seg", "MedMNIST": "cls", "CIFAR100": "cls", "AG_NEWS": "cls", "MTSamples": "cls",
                  "MedicalAbstracts": "cls", "Flickr30k": "rtv", "Coco": "rtv"}
DATASET_2_MODALITY = {"BraTS": "t1", "MedMNIST": "img", "CIFAR100": "img", "AG_NEWS": "txt", "MTSamples": "txt",
                      "MedicalAbstracts": "txt", "Flickr30k": "img+txt", "Coco": "img+txt"}
NUM_CLASS = {"CIFAR100": 100, "AG_NEWS": 4, "MedMNIST": 11, "MTSamples": 40, "MedicalAbstracts": 5, "Flickr30k": None, "Coco": None}
TASK_2_CRITERION = {"cls": "CrossEntropyLoss", "seg": "SegLoss", "img+txt": "ContrastiveLoss"}
VOCAB_SIZES = {"Flickr30k": 7732, "MedicalAbstracts": 20264}
MM_METRICS = ("recall_1", "recall_5", "recall_10", "rsum")

get_name_type = agg.get_name_type
get_name_modality = agg.get_name_modality


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


class FedavgServer(BaseServer):
    def __init__(self, args, writer, server_dataset, client_datasets, model_str):
        super().__init__()
        self.args = args
        self.writer = writer
        self.round = 0
        self.server_dataset = server_dataset[1] if (args.eval_type != "local" and server_dataset is not None) else None
        self.global_models = self._init_model(model_str)
        self._init_param_scope(args.shared_param, args.share_scope)
        self._set_evaluator()
        self.opt_kwargs = dict(lr=self.args.lr, momentum=getattr(self.args, "beta1", 0.0))
        self.curr_lr = self.args.lr
        self.clients = self._create_clients(client_datasets)
        self.results = defaultdict(dict)
        self.server_device = self.args.server_device
        if type(args.Cs) != list or len(args.Cs) == 1:                                 # fedavgserver.py:137-141
            self.args.Cs = (self.args.Cs * len(self.args.datasets)) if (type(args.Cs) == list) else [self.args.Cs] * len(self.args.datasets)
        self.Cs = {dataset: C for dataset, C in zip(self.args.datasets, self.args.Cs)}

    # ------------------------------------------------------------------ construction
    def _init_model(self, model_str):
        """fedavgserver.py:144-158 (note: drops the last entry of args.datasets, like the reference)."""
        self.args.datasets = self.args.datasets[:-1]
        models = {}
        a = self.args
        for i, dataset in enumerate(a.datasets):
            a.vocab_size = VOCAB_SIZES[dataset] if dataset in VOCAB_SIZES else 30522
            kw = dict(pretrained=a.pretrained, args=a, with_aux=a.with_aux, aux_trained=a.aux_trained, aux_attn_only=a.aux_attn_only,
                      aux_mlp_only=a.aux_mlp_only)
            mod = DATASET_2_MODALITY[dataset]
            if mod == "img":
                models[dataset] = create_model(model_str, num_classes=[NUM_CLASS[dataset], None], modalities=[a.modalities[i], None],
                                               tasks=[DATASET_2_TASK[dataset], None], **kw)
            elif mod == "txt":
                models[dataset] = create_model(model_str, num_classes=[None, NUM_CLASS[dataset]], modalities=[None, a.modalities[i]],
                                               tasks=[None, DATASET_2_TASK[dataset]], **kw)
            elif mod == "img+txt":
                models[dataset] = create_model(model_str, num_classes=[None, None], modalities=["img", "txt"],
                                               tasks=[DATASET_2_TASK[dataset], DATASET_2_TASK[dataset]], **kw)
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        for m in models.values():
            m.to(dev)
        return models

    def _init_param_scope(self, shared_param, share_scope):
        names = []
        for model in self.global_models.values():
            for key in model.state_dict().keys():
                if key not in names:
                    names.append(key)
        self.param_scope = agg.init_param_scope(names, shared_param, share_scope)

    def sync_shared_params(self):
        """fedavgserver.py:160-168 (no caller in the reference): every key whose scope is not 'dataset' is copied from the LAST dataset's model."""
        sd = self.global_models[self.args.datasets[-1]].state_dict()
        for model in self.global_models.values():
            new_sd = model.required_params()
            for k, v in sd.items():
                if k in new_sd.keys() and self.param_scope[k] != "dataset":
                    new_sd[k] = v
            model.load_state_dict(new_sd, strict=False)

    def _set_loaders(self, datasets):
        self.server_dataset = datasets[1]                                    # fedavgserver.py:170-171

    def _refine_optim_args(self, args):
        """fedavgserver.py:432-440."""
        import inspect
        required_args = inspect.getfullargspec(torch.optim.__dict__[self.args.optimizer])[0]
        return {a: getattr(args, a) for a in required_args if hasattr(args, a)}

    def _get_algorithm(self, model, **kwargs):
        """fedavgserver.py:241-246: the (dormant) src/algorithm plugin point."""
        cls = import_module(f"..algorithm.{self.args.algorithm}", package=__package__).__dict__[f"{self.args.algorithm.title()}Optimizer"]
        return cls(params=model.state_dict(), **kwargs)

    def _create_clients(self, client_datasets):
        """fedavgserver.py:248-280 (plugin lookup f'{algorithm}client'.{Algorithm}Client kept)."""
        cls = import_module(f"..client.{self.args.algorithm}client", package=__package__).__dict__[f"{self.args.algorithm.title()}Client"]
        clients = []
        for identifier, datasets in enumerate(client_datasets):
            client = cls(args=self.args, training_set=datasets[0], test_set=datasets[1], task=datasets[2], modality=datasets[3],
                         eval_metrics=["acc1"] if datasets[2] == "cls" else ["f1"], criterion=TASK_2_CRITERION[datasets[2]]
                         if datasets[3] != "img+txt" else TASK_2_CRITERION["img+txt"], writer=self.writer)
            client.id = identifier
            client.dataset = datasets[4]
            client.device = "cuda" if torch.cuda.is_available() else "cpu"
            clients.append(client)
        clients.sort(key=lambda c: c.id)
        return clients

    # ------------------------------------------------------------------ sampling (bit-exact with the reference)
    def _sample_clients(self, exclude=[]):
        """fedavgserver.py:282-312.  Uses Python ``random`` (seeded by utils.set_seed) in the reference's call order."""
        if self.args.equal_sampled:
            sampled = []
            for i, dataset in enumerate(self.args.datasets):
                ids = [client.id for client in self.clients if client.dataset == dataset]
                n = max(int(self.Cs[dataset] * len(ids)), 1)
                sampled += sorted(random.sample(ids, n))
            sampled = sorted(sampled)
        else:
            if exclude == []:
                n = max(int(self.args.C * self.args.K), 1)
                sampled = sorted(random.sample([i for i in range(self.args.K)], n))
            else:
                rest = self.args.K - len(exclude)
                if rest == 0:
                    sampled = sorted([i for i in range(self.args.K)])
                else:
                    n = max(int(self.args.eval_fraction * rest), 1)
                    sampled = sorted(random.sample([i for i in range(self.args.K) if i not in exclude], n))
        if self.args.warmup_modality != "none" and self.round <= self.args.warmup_rounds:
            sampled = [i for i in sampled if self.clients[i].modality == self.args.warmup_modality]
        return sampled

    def _owner_rank(self, position: int, world: int) -> int:
        return position % world                                           # reference: cuda:(i % n_gpu) by position (:310-311)

    # ------------------------------------------------------------------ request fan-out
    def _freeze_shared_params(self, client):
        for name in client.model.segments:
            if self.param_scope.get(name) == "all":
                client.model.set_trainable(name, False)

    def _unfreeze_params(self, client):
        """fedavgserver.py:426-429: requires_grad = True on EVERY parameter -- including ``aux_weight`` of a model built with
        ``aux_trained=False``: past the freeze window the freeze-modality clients train their aux weights (a quirk of the reference that
        its results contain, so it is kept)."""
        for name in client.model.segments:
            client.model.set_trainable(name, True)

    def _request(self, ids, eval=False, participated=True, retain_model=True, save_raw=False):
        """fedavgserver.py:505-589 (update path).  Each rank trains the sampled clients it owns; sizes/results are exchanged so
        that every rank has the same ``updated_sizes``."""
        dist, rank, world = _dist()
        if eval:                                                             # fedavgserver.py:522-555: clients' hold-out evaluation
            if self.args.train_only:
                return None
            sizes, results = {}, {}
            for pos, idx in enumerate(ids):
                if self._owner_rank(pos, world) != rank:
                    continue
                client = self.clients[idx]
                client.download(self.global_models)                          # (require_model=True: always the current global model)
                results[client.id] = client.evaluate()
                sizes[client.id] = len(client.test_set)
                if not retain_model:
                    client.model = None
            if world > 1:
                gathered = [None] * world
                dist.all_gather_object(gathered, (sizes, results))
                sizes, results = {}, {}
                for s_, r_ in gathered:
                    sizes.update(s_)
                    results.update(r_)
            self.results[self.round][f'clients_evaluated_{"in" if participated else "out"}'] = {str(k): v for k, v in results.items()}
            return None
        sizes, results = {}, {}
        for pos, idx in enumerate(ids):
            if self._owner_rank(pos, world) != rank:
                continue
            client = self.clients[idx]
            if client.model is None:
                client.download(self.global_models)
            client.args.lr = self.curr_lr                                    # fedavgserver.py:509
            if self.args.freeze_modality != "none" and client.modality == self.args.freeze_modality:
                if self.args.warmup_rounds < self.round <= (self.args.freeze_rounds + self.args.warmup_rounds):
                    self._freeze_shared_params(client)
                elif self.round > (self.args.freeze_rounds + self.args.warmup_rounds):
                    self._unfreeze_params(client)
            self._before_client_update(client)
            results[client.id] = client.update()
            self._after_client_update(client)
            sizes[client.id] = len(client.training_set)
            if not retain_model:
                client.model = None
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, (sizes, results))
            sizes, results = {}, {}
            for s, r in gathered:
                sizes.update(s)
                results.update(r)
        self.results[self.round]["clients_updated"] = {str(k): v for k, v in results.items()}
        return sizes

    def _before_client_update(self, client):
        """Hook of the update fan-out (no-op here; CreamFL hands the global public features to the client)."""

    def _after_client_update(self, client):
        """Hook of the update fan-out (no-op here; CreamFL refreshes the client's public features)."""

    # ------------------------------------------------------------------ aggregation hook
    def _client_upload_segments(self, client):
        """Keys (and their offsets in the client's flat buffer) that ``client.upload()`` would return, from bookkeeping alone."""
        model = client.model if client.model is not None else self.global_models[client.dataset]
        drop_aux = self.args.with_aux and client.modality != "img+txt"
        segs = {k: s for k, s in model.segments.items() if not (drop_aux and ("aux" in k or "cross_modal_scale" in k))}
        for ak, tk in model._alias_keys():       # shared tensors (colearn_param 'attn', scope 'all'): the upload lists them under both keys
            if tk in segs:
                segs[ak] = segs[tk]
        return segs

    def _aggregate_plan(self, ids, updated_sizes, fedavg=False):
        """Host half of _aggregate for the current self.global_model / task / modality / dataset: the reference's coefficient table, the
        blend plan and the flat buffers of the sampled clients this rank trained."""
        assert set(updated_sizes.keys()) == set(ids)
        keys = list(self.global_model.required_params().keys())
        coefficients = agg.mixing_coefficients(keys, self.param_scope, updated_sizes, self.clients, dataset=self.dataset, task=self.task,
                                               modality=self.modality, out_modality_scale=self.out_modality_scale, args=self.args,
                                               fedavg=fedavg)
        client_segments = {i: self._client_upload_segments(self.clients[i]) for i in ids}
        plan = agg.build_plan(self.global_model, ids, coefficients, client_segments)
        local_flats = {}
        for i in ids:
            c = self.clients[i]
            if c.model is None:
                continue                                                      # trained on another rank
            c.upload()                                                        # folds aux into the weights when needed
            local_flats[i] = getattr(c, "_folded", None) if (self.args.with_aux and c.modality != "img+txt") else c.model.flat.data
        return plan, local_flats, keys, coefficients, client_segments

    def _aggregate(self, ids, updated_sizes, fedavg=False, local_partial=None, all_reduce=None, exact=False):
        """fedavgserver.py:591-668 with the same inputs (self.global_model / task / modality / dataset / out_modality_scale /
        param_scope / clients).  ``self.comm`` (fedcola_amd.comm.Comm, optional) routes the cross-rank sum through the C ABI's own
        RCCL communicator instead of torch.distributed.  ``exact=True``: the reference's sequential loop itself on the device
        (bit-identical rounding; verification mode -- every sampled client local, or one client per rank with ``self.comm``)."""
        plan, local_flats, keys, coefficients, client_segments = self._aggregate_plan(ids, updated_sizes, fedavg)
        dist, rank, world = _dist()
        comm = getattr(self, "comm", None)
        if exact:
            agg.aggregate_exact(self.global_model, keys, ids, coefficients, client_segments, local_flats, comm=comm)
            return
        kw = {} if local_partial is None else {"local_partial": local_partial}
        agg.aggregate(self.global_model, plan, local_flats, rank=rank, world=world, all_reduce=all_reduce, comm=comm, **kw)

    def _empty_client_models(self):
        for client in self.clients:
            client.model = None
        gc.collect()

    def _set_evaluator(self):
        """fedavgserver.py:177-181"""
        from ..metrics.eval_coco import COCOEvaluator
        evaluator = COCOEvaluator("matmul", n_crossfolds=5, extract_device=self.args.server_device, eval_device=self.args.server_device,
                                  verbose=False)
        evaluator.set_logger(logger)
        self.evaluator = evaluator

    def _eval_loader(self, dataset, batch_size, shuffle):
        # the reference spawns 4 persistent loader workers (fedavgserver.py:687,725); loader plumbing is the caller's
        return torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=shuffle,
                                           num_workers=getattr(self.args, "eval_num_workers", 0))

    @torch.no_grad()
    def _central_evaluate(self, fedavg=False):
        """fedavgserver.py:676-760: img+txt datasets through the retrieval evaluator (1k 5-fold + full-gallery recall), uni-modal
        datasets through a forward / criterion / MetricManager loop on the server device."""
        from ..criterions import CRITERIA
        from ..utils import MetricManager
        out = {}
        for dataset in self.server_dataset.keys():
            tag = f"[{self.args.algorithm.upper()}] [{dataset.upper()}] [Round: {str(self.round).zfill(4)}] [EVALUATE] [SERVER] "
            if DATASET_2_MODALITY[dataset] == "img+txt":
                self.global_model = self.global_models[dataset]
                self.evaluator.set_model(self.global_model)
                kw = {k: getattr(self.args, k) for k in ("n_images_per_crossfold", "n_captions_per_crossfold") if hasattr(self.args, k)}
                result = self.evaluator.evaluate(self._eval_loader(self.server_dataset[dataset], self.args.eval_batch_size, True),
                                                 eval_batch_size=self.args.eval_batch_size, **kw)
                res_dict = {}
                for fold, src in (("1k", result.get("n_fold")), ("5k", result)):
                    if src is None:
                        continue
                    for task in ("i2t", "t2i"):
                        for metric in MM_METRICS:
                            tag += f"| {dataset}{fold}_{task}_{metric}: {src[task][metric]:.4f} "
                            res_dict[f"Result/Server {dataset} {fold}_{task}_{metric.title()}"] = src[task][metric]
                    r1 = src["t2i"]["recall_1"] + src["i2t"]["recall_1"]
                    tag += f"| {dataset} {fold}_rsum: {r1:.4f} "
                    res_dict[f"Test/Server {dataset} {fold}_r@1sum"] = r1
                if result.get("n_fold") is not None:
                    res_dict[f"Test/Server {dataset} r@1sum"] = (res_dict[f"Test/Server {dataset} 1k_r@1sum"]
                                                                 + res_dict[f"Test/Server {dataset} 5k_r@1sum"])
                logger.info(tag)
                if self.writer is not None:
                    self.writer.log(res_dict, self.round)
                out[dataset] = result
            else:
                self.global_model = self.global_models[dataset]
                mm = MetricManager(self.args.eval_metrics)
                self.global_model.eval()
                self.global_model.to(self.args.server_device)
                n = 0
                for inputs, targets in self._eval_loader(self.server_dataset[dataset], self.args.B, False):
                    inputs, targets = inputs.to(self.args.server_device), targets.to(self.args.server_device)
                    if DATASET_2_MODALITY[dataset] == "img":
                        outputs = self.global_model([inputs, None])[0]
                    else:
                        outputs = self.global_model([None, inputs])[1]
                    loss = CRITERIA[self.args.criterion]()(outputs, targets)
                    mm.track(loss.item(), outputs, targets)
                mm.aggregate(len(self.server_dataset[dataset]))
                result = mm.results
                tag += f"| loss: {result['loss']:.4f} " + "".join(f"| {k}: {v:.4f} " for k, v in result["metrics"].items())
                logger.info(tag)
                suffix = dataset + ("after" if not fedavg else "")
                if self.writer is not None:
                    self.writer.log({f"Loss/Server {suffix} Loss": result["loss"]}, self.round)
                    for name, value in result["metrics"].items():
                        self.writer.log({f"Test/Server {suffix} {name.title()}": value}, self.round)
                self.results[self.round][f"server_evaluated_{suffix}"] = result
                out[dataset] = result
        return out

    # ------------------------------------------------------------------ one federated round (fedavgserver.py:784-856)
    def update(self):
        selected_ids = self._sample_clients()
        updated_sizes = self._request(selected_ids, eval=False, participated=True, retain_model=True, save_raw=False)
        _, rank, world = _dist()
        if getattr(self.args, "fedavg_eval", False):                          # fedavgserver.py:796-808: what plain FedAvg would have produced, evaluated
            import copy                                                       # centrally, then thrown away -- the round continues from the old models
            old_models = copy.deepcopy(self.global_models)
            for i, dataset in enumerate(self.global_models.keys()):
                self.global_model = self.global_models[dataset]
                self.task = DATASET_2_TASK[dataset]
                self.modality = DATASET_2_MODALITY[dataset]
                self.dataset = dataset
                self.out_modality_scale = self.args.out_modality_scales[i]
                self._aggregate(selected_ids, updated_sizes, fedavg=True)
                self.global_models[dataset] = self.global_model
            self._central_evaluate(fedavg=True)
            self.global_models = old_models
        items = []
        for i, dataset in enumerate(self.global_models.keys()):
            self.global_model = self.global_models[dataset]
            self.task = DATASET_2_TASK[dataset]
            self.modality = DATASET_2_MODALITY[dataset]
            self.dataset = dataset
            self.out_modality_scale = self.args.out_modality_scales[i]
            if world > 1 and self.global_model.flat.is_cuda and type(self)._aggregate is FedavgServer._aggregate:
                # several ranks: every dataset's global model of this round goes through ONE all-reduce (aggregate_many)
                plan, local_flats, *_ = self._aggregate_plan(selected_ids, updated_sizes)
                items.append((self.global_model, plan, local_flats))
            else:
                self._aggregate(selected_ids, updated_sizes)
            self.global_models[dataset] = self.global_model
        if items:
            agg.aggregate_many(items, rank=rank, world=world, comm=getattr(self, "comm", None))
        if self.args.with_aux:                                                # fedavgserver.py:821-845
            for dataset in self.global_models.keys():
                gm = self.global_models[dataset]
                modality = DATASET_2_MODALITY[dataset]
                if modality == "img+txt":
                    continue
                other = "txt" if modality == "img" else "img"
                src_ds = [d for d in self.global_models.keys() if DATASET_2_MODALITY[d] == other][0]
                sd = self.global_models[src_ds].state_dict()
                a, b = ("blockses.0", "blockses.1") if modality == "img" else ("blockses.1", "blockses.0")
                auxes = {k: sd[k.replace("aux_", "").replace(a, b)] for k in gm.aux_params().keys()}
                gm.load_state_dict(auxes, strict=False)
        if self.round % self.args.lr_decay_step == 0:
            self.curr_lr *= self.args.lr_decay
        self._empty_client_models()
        return selected_ids

    def evaluate(self, excluded_ids):
        """fedavgserver.py:858-869: 'local' / 'both' evaluate every client's hold-out set, 'global' / 'both' the server's."""
        if self.args.eval_type != "global":
            self._request(range(self.args.K), eval=True, participated=False, retain_model=False, save_raw=self.round == getattr(self.args, "R", -1))
        if self.args.eval_type != "local":
            self._central_evaluate()

    def finalize(self):
        """fedavgserver.py:884-895: results json + one state_dict checkpoint per dataset (reference key names)."""
        os.makedirs(self.args.result_path, exist_ok=True)
        with open(os.path.join(self.args.result_path, f"{self.args.exp_name}.json"), "w", encoding="utf8") as f:
            json.dump({str(k): v for k, v in self.results.items()}, f, indent=4, default=str)
        d = os.path.join(self.args.result_path, f"{self.args.exp_name}")
        os.makedirs(d, exist_ok=True)
        for dataset, model in self.global_models.items():
            torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, os.path.join(d, f"{dataset}.pt"))
