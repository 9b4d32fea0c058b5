"""``CreamflServer`` (/root/reference/src/server/creamflserver.py:26-435): FedAvg plumbing plus CreamFL's server half.

Per round (``update``, creamflserver.py:338-435):
  1. ``_generate_public_logit``: the img+txt global model's features of the public set (kept on the device).
  2. clients update with those features (``CreamflClient``), uni-modal clients then refresh their own public features.
  3. ``aggregation``: per client log-prob diagonal against the other modality's global features, softmax over clients,
     weighted feature sum                                                  -> fc_cream_logprob_diag + fc_cream_combine
  4. img+txt global model: zero-initialised weighted sum of the uploads (NOT the sequential blend)  -> fc_aggregate_blend with w_g = 0,
     then KD distillation to the aggregated features: MSE, clip 2, AdamW(p_lr) -> fc_forward / fc_mse_loss_fwd_bwd / fc_backward /
     fc_clip_grad_norm / fc_adamw_step
  5. uni-modal global models: ``FedavgServer._aggregate(fedavg=True)``.
The public set is ``args.pub_dataset`` when given (samples ``(image, tokens, image_id, ann_id, index)``), else the COCO split the reference
builds (creamflserver.py:100-126) through ``fedcola_amd.datasets.coco.public_set`` (image transform / tokenizer: ``args.pub_transform``,
``args.pub_tokenizer`` -- the reference hard-wires torchvision transforms and the HF BertTokenizer there)."""
from __future__ import annotations

import logging
import operator

import torch
from torch.utils import data

from .. import _lib, aggregate as agg
from .._lib import check, ptr
from .fedavgserver import DATASET_2_MODALITY, DATASET_2_TASK, FedavgServer, _dist

logger = logging.getLogger(__name__)


class CreamflServer(FedavgServer):
    def __init__(self, args, writer, server_dataset, client_datasets, model_str):
        pub = getattr(args, "pub_dataset", None)
        if pub is None:                                           # creamflserver.py:100-113: the COCO public split
            from ..datasets.coco import public_set
            pub = public_set(args.pub_data_dir, args.pub_anno_path, args.num_pub_samples, transform=getattr(args, "pub_transform", None),
                             tokenizer=getattr(args, "pub_tokenizer", None), max_length=args.seq_len)
        self.pub_dataset = pub
        self.pub_loader = data.DataLoader(dataset=pub, batch_size=args.pub_batch_size, shuffle=False, drop_last=False)
        super().__init__(args, writer, server_dataset, client_datasets, model_str)
        self.device = args.server_device

    def _create_clients(self, client_datasets):
        clients = super()._create_clients(client_datasets)
        for c in clients:
            c.pub_dataset = self.pub_dataset                      # creamflserver.py:48 (deepcopy there; read-only here)
        return clients

    # ------------------------------------------------------------------ public features of the global model (:128-162)
    @torch.no_grad()
    def _generate_public_logit(self):
        model = [m for ds, m in self.global_models.items() if DATASET_2_MODALITY[ds] == "img+txt"][-1]
        model.eval()
        model.to(self.device)
        fi, ft, index = [], [], []
        for images, captions, _, _, idx in self.pub_loader:
            outs = model([images.to(self.device), captions.to(self.device)])
            fi.append(outs[0].detach().float().clone())
            ft.append(outs[1].detach().float().clone())
            index.extend(idx)
        self.global_img_feature = torch.cat(fi, dim=0)
        self.global_txt_feature = torch.cat(ft, dim=0)
        self.distill_index = index

    def _before_client_update(self, client):
        client.global_img_feature = self.global_img_feature      # creamflserver.py:171-176
        client.global_txt_feature = self.global_txt_feature
        client.distill_index = self.distill_index

    def _after_client_update(self, client):
        if client.modality != "img+txt":                          # creamflserver.py:180-181
            client.update_pub_feature()

    # ------------------------------------------------------------------ feature aggregation (:372-407)
    def aggregate_features(self, vecs, g_other):
        if len(vecs) == 0:
            return None
        L = _lib.lib()
        dev = torch.device(self.device)
        vs = [v.to(dev).float().contiguous() for v in vecs]
        g = g_other.to(dev).float().contiguous()
        P, D = vs[0].shape
        w = torch.empty(len(vs), P, device=dev)
        sp = _lib.stream_ptr()
        for c, v in enumerate(vs):
            check(L.fc_cream_logprob_diag(ptr(v), ptr(g), P, D, ptr(w[c]), sp))
        table = torch.tensor([v.data_ptr() for v in vs], dtype=torch.int64, device=dev)
        out = torch.empty(P, D, device=dev)
        check(L.fc_cream_combine(ptr(table), ptr(w), len(vs), P, D, ptr(out), sp))
        torch.cuda.current_stream().synchronize()
        return out

    # ------------------------------------------------------------------ img+txt global model (:251-336)
    def _aggregate(self, ids, updated_sizes, fedavg=False, **kw):
        if fedavg:
            return super()._aggregate(ids, updated_sizes, fedavg=True, **kw)
        assert set(updated_sizes.keys()) == set(ids)
        gm = self.global_model
        keys = list(gm.state_dict().keys())
        keys = [k for k in keys if k in gm.segments]
        coefficients = {}
        for k in keys:                                            # creamflserver.py:261-277 (plain coefficients, strict modality test)
            num = {}
            for i, n in updated_sizes.items():
                c, sc = self.clients[i], self.param_scope[k]
                num[i] = n if (sc == "all" or (sc == "dataset" and c.dataset == self.dataset) or (sc == "task" and c.task == self.task)
                               or (sc == "modality" and c.modality == self.modality)) else 0
            den = sum(updated_sizes.values()) if self.args.compensation else sum(num.values())
            coefficients[k] = {i: float(v / den) for i, v in num.items()}
        dist, rank, world = _dist()
        segs = {i: self._client_upload_segments(self.clients[i]) for i in ids}
        plan = agg.build_plan(gm, ids, coefficients, segs, zero_init=True)
        flats = {i: self.clients[i].model.flat.data for i in ids if self.clients[i].model is not None}
        agg.aggregate(gm, plan, flats, rank=rank, world=world)
        self._kd_distill()

    def _kd_distill(self):
        """creamflserver.py:290-336"""
        L = _lib.lib()
        gm, args = self.global_model, self.args
        dev = torch.device(self.device)
        gm.train()
        gm.to(dev)
        n = gm.flat.numel()
        grads, m1, m2 = (torch.zeros(n, device=dev) for _ in range(3))
        lossbuf = torch.zeros(2, device=dev)
        clip_scratch = torch.empty(L.fc_clip_scratch_bytes(gm._handle.h), dtype=torch.uint8, device=dev)
        distill_dict = {int(b): a for a, b in enumerate(self.distill_index)}
        D = gm.embed_dim
        sp = _lib.stream_ptr()
        img_vec, txt_vec = self.img_vec.to(dev).float().contiguous(), self.txt_vec.to(dev).float().contiguous()
        for step, (images, captions, _, _, index) in enumerate(self.pub_loader, 1):
            B = images.shape[0]
            img, ids = images.to(dev).contiguous().float(), captions.to(dev).contiguous().long()
            d_idx = torch.tensor(list(operator.itemgetter(*index.tolist())(distill_dict)) if B > 1 else [distill_dict[int(index[0])]],
                                 dtype=torch.int64, device=dev)
            gm.prepare_weights()
            ws = gm.workspace(B, ids.shape[1])
            dp = gm.make_droppath(B)
            oi, ot = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
            check(L.fc_forward(gm._handle.h, ptr(gm.flat), ptr(gm._wc_or_flat()), ptr(img), ptr(ids), B, ids.shape[1], 0, ptr(dp), ptr(ws),
                               ws.numel(), ptr(oi), ptr(ot), sp))
            ti, tt = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
            di, dt = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
            check(L.fc_gather_rows(ptr(img_vec), ptr(d_idx), B, D, ptr(ti), sp))
            check(L.fc_gather_rows(ptr(txt_vec), ptr(d_idx), B, D, ptr(tt), sp))
            check(L.fc_mse_loss_fwd_bwd(ptr(oi), ptr(ti), B * D, float(args.kd_weight), B, ptr(lossbuf), ptr(di), sp))
            check(L.fc_mse_loss_fwd_bwd(ptr(ot), ptr(tt), B * D, float(args.kd_weight), B, ptr(lossbuf), ptr(dt), sp))
            grads.zero_()
            check(L.fc_backward(gm._handle.h, ptr(gm.flat), ptr(gm._wc_or_flat()), ptr(di), ptr(dt), ptr(grads), ptr(ws), ws.numel(), sp))
            check(L.fc_clip_grad_norm(gm._handle.h, ptr(grads), 2.0, ptr(clip_scratch), clip_scratch.numel(), None, sp))
            check(L.fc_adamw_step(gm._handle.h, ptr(gm.flat), ptr(grads), ptr(m1), ptr(m2), float(args.p_lr), 0.9, 0.999, 1e-8, 0.01, step, sp))
            gm._bump()
            torch.cuda.current_stream().synchronize()

    def _exchange_pub_features(self, selected_ids):
        """One process per GPU: a uni-modal client's public-set features exist only on the rank that trained it this round, but
        every rank runs the (replicated) feature aggregation and distillation of creamflserver.py:373-435 and must see the same
        inputs.  Each owner's [P, D] features are all-gathered in selected-id order; a rank that still holds stale features of a
        client it owned in an earlier round gets them overwritten.  Single process: nothing to do."""
        from .fedavgserver import _dist
        dist, rank, world = _dist()
        if world == 1:
            return
        mine = {}
        for pos, i in enumerate(selected_ids):
            c = self.clients[i]
            if c.modality in ("img", "txt") and self._owner_rank(pos, world) == rank:
                mine[i] = c.pub_features.detach().float().cpu()
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        dev = torch.device(self.device)
        for pos, i in enumerate(selected_ids):
            c = self.clients[i]
            if c.modality not in ("img", "txt"):
                continue
            owner = self._owner_rank(pos, world)
            if i not in gathered[owner]:
                raise RuntimeError(f"client {i}: rank {owner} trained it but sent no public-set features")
            if owner != rank:
                c.pub_features = gathered[owner][i].to(dev)

    # ------------------------------------------------------------------ one round (:338-435)
    def update(self):
        self._generate_public_logit()
        selected_ids = self._sample_clients()
        updated_sizes = self._request(selected_ids, eval=False, participated=True, retain_model=True, save_raw=False)
        self._exchange_pub_features(selected_ids)
        img_vec, txt_vec = [], []
        for i in selected_ids:
            c = self.clients[i]
            if c.modality == "img":
                img_vec.append(c.pub_features)
            elif c.modality == "txt":
                txt_vec.append(c.pub_features)
        self.img_vec = self.aggregate_features(img_vec, self.global_txt_feature) if img_vec else []
        self.txt_vec = self.aggregate_features(txt_vec, self.global_img_feature)
        for dataset in self.global_models.keys():
            self.global_model = self.global_models[dataset]
            self.task, self.modality, self.dataset = DATASET_2_TASK[dataset], DATASET_2_MODALITY[dataset], dataset
            if self.modality == "img+txt":
                self._aggregate(selected_ids, updated_sizes)
            else:
                self.out_modality_scale = 1
                super()._aggregate(selected_ids, updated_sizes, fedavg=True)
            self.global_models[dataset] = self.global_model
        if self.round % self.args.lr_decay_step == 0:
            self.curr_lr *= self.args.lr_decay
        self._empty_client_models()
        return selected_ids
