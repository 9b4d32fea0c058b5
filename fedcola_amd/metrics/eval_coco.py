"""Retrieval evaluation: mirror of the reference's ``COCOEvaluator`` (src/metrics/eval_coco.py:89-469, method 'matmul',
the only one FedavgServer builds, src/server/fedavgserver.py:177-181) with the similarity / ranking work on the GPU.

Same surface (``set_model, set_criterion, set_logger, extract_features, retrieve, evaluate_recall, evaluate_n_fold,
evaluate``), same result dictionaries.  What changed underneath:
  * extract_features: the reference copies every sample's feature to numpy inside a Python loop (eval_coco.py:184-198);
    here the batch outputs stay on the device, are concatenated once, and the de-duplication / regrouping bookkeeping is
    vectorised (``collect``).
  * evaluate_recall: the reference sorts every similarity row and searches it with ``torch.where`` per positive
    (eval_coco.py:327-334); here ``fc_retrieval_best_ranks`` (fp64 MFMA similarity GEMM + one counting pass per query,
    include/fedcola_hip.h) returns the same integer ranks.
There is no CPU fallback: without the HIP library / a GPU the ranking raises.
"""
from __future__ import annotations

from functools import partial

import numpy as np
import torch

from .. import _lib


def recall_at_k(ranks, k):
    """eval_coco.py:38-45"""
    return 100.0 * len(np.where(ranks < k)[0]) / len(ranks)


def collect(img_f: torch.Tensor, cap_f: torch.Tensor, image_ids: torch.Tensor, ann_ids: torch.Tensor, num_images: int, num_captions: int,
            iid_to_cls):
    """Bookkeeping of extract_features (eval_coco.py:148-248) over the concatenated loader stream.

    img_f / cap_f: [n, D] features in loader order (one row per caption sample); image_ids / ann_ids: [n].
    Images are kept at their first occurrence; without a class map the captions are regrouped to follow the image order."""
    ids = image_ids.detach().cpu().numpy().astype(np.int64)
    anns = ann_ids.detach().cpu().numpy().astype(np.float64)
    n, D = cap_f.shape
    uniq, first = np.unique(ids, return_index=True)
    order = np.argsort(first, kind="stable")
    first_sorted = first[order]                       # sample index of each image's first occurrence, in first-seen order
    image_ids_ = uniq[order].astype(np.float64)
    if len(image_ids_) != num_images:
        raise RuntimeError("unexpected error, {} != {}".format(len(image_ids_), num_images))
    if n != num_captions:
        raise RuntimeError("unexpected error, {}, {}".format(n, num_captions))
    if iid_to_cls:
        cls_of = np.vectorize(lambda i: iid_to_cls.get(int(i), int(i)), otypes=[np.float64])
        image_classes = cls_of(uniq[order])
        caption_classes = cls_of(ids)
    else:
        image_classes = image_ids_.copy()
        caption_classes = ids.astype(np.float64)
    if set(image_classes) != set(caption_classes):
        raise RuntimeError("unexpected error, I({}) != C({})".format(set(image_classes), set(caption_classes)))
    image_features = img_f[torch.as_tensor(first_sorted, device=img_f.device)].to(torch.float64).reshape(num_images, 1, D)
    caption_features = cap_f.to(torch.float64).reshape(n, 1, D)
    caption_ids = anns
    if not iid_to_cls:
        # captions follow the image order; within an image, loader order (np.where in the reference is ascending)
        rank_of = {v: i for i, v in enumerate(image_classes)}
        key = np.fromiter((rank_of[c] for c in caption_classes), dtype=np.int64, count=n)
        sorted_caption_idx = np.argsort(key, kind="stable")
        caption_ids = caption_ids[sorted_caption_idx]
        caption_classes = caption_classes[sorted_caption_idx]
        caption_features = caption_features[torch.as_tensor(sorted_caption_idx, device=caption_features.device)]
    return {
        "image_features": image_features.cpu(),
        "caption_features": caption_features.cpu(),
        "image_sigmas": np.zeros((num_images, D)),
        "caption_sigmas": np.zeros((n, D)),
        "image_ids": image_ids_,
        "caption_ids": caption_ids,
        "image_classes": torch.from_numpy(image_classes),
        "caption_classes": torch.from_numpy(caption_classes),
    }


def _labels_i64(x, device):
    t = torch.as_tensor(np.asarray(x)) if not torch.is_tensor(x) else x
    return t.to(torch.float64).round().to(torch.int64).to(device).contiguous()


def best_ranks_device(q_features, g_features, q_labels, g_labels, device="cuda", batch_size=1024):
    """best_pred_ranks of evaluate_recall (eval_coco.py:320-334) through the C ABI; returns a float64 numpy array."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.FedcolaHipError("fedcola_amd retrieval evaluation runs on the GPU only (eval_device must be a cuda device)")
    L = _lib.lib()
    nq, ng = len(q_labels), len(g_labels)
    q = torch.as_tensor(q_features).to(dev, torch.float64).reshape(nq, -1).contiguous()
    g = torch.as_tensor(g_features).to(dev, torch.float64).reshape(ng, -1).contiguous()
    if q.shape[1] != g.shape[1]:
        raise RuntimeError("feature size mismatch {}, {}".format(tuple(q.shape), tuple(g.shape)))
    ql, gl = _labels_i64(q_labels, dev), _labels_i64(g_labels, dev)
    qb = max(1, min(int(batch_size), nq))
    scratch = torch.empty(L.fc_retrieval_scratch_bytes(qb, ng), dtype=torch.uint8, device=dev)
    out = torch.empty(nq, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.fc_retrieval_best_ranks(_lib.ptr(q), _lib.ptr(g), _lib.ptr(ql), _lib.ptr(gl), nq, ng, q.shape[1], _lib.ptr(scratch),
                                             scratch.numel(), _lib.ptr(out), _lib.stream_ptr()))
    ranks = out.cpu().numpy()
    if (ranks < 0).any():
        raise ValueError("min() arg is an empty sequence")          # the reference's failure for a query without positives
    return ranks.astype(np.float64)


class COCOEvaluator(object):
    """eval_coco.py:89-110"""

    def __init__(self, eval_method="matmul", n_crossfolds=-1, extract_device="cuda", eval_device="cuda", verbose=False):
        if eval_method != "matmul":
            raise NotImplementedError("fedcola_amd COCOEvaluator implements eval_method='matmul' (the one FedavgServer builds)")
        self.eval_method = eval_method
        self.extract_device = extract_device
        self.eval_device = eval_device
        self.logger = None
        self.n_crossfolds = n_crossfolds
        try:
            from tqdm import tqdm
            self.pbar = partial(tqdm, disable=not verbose)
        except ImportError:  # pragma: no cover
            self.pbar = lambda it, **_k: it

    def set_model(self, model):
        self.model = model
        self.n_embeddings = 1
        self.feat_size = self.model.embed_dim

    def set_criterion(self, criterion):
        self.criterion = criterion

    def set_logger(self, logger):
        self.logger = logger

    @torch.no_grad()
    def extract_features(self, dataloader):
        """eval_coco.py:134-248"""
        self.model.eval()
        self.model.to(self.extract_device)
        num_images = dataloader.dataset.n_images
        num_captions = len(dataloader.dataset)
        iid_to_cls = dataloader.dataset.iid_to_cls
        fi, fc, ii, ai = [], [], [], []
        for images, captions, image_ids, ann_ids, _ in self.pbar(dataloader):
            images = images.to(self.extract_device)
            captions = captions.to(self.extract_device)
            output = self.model([images, captions], feat_out=True)
            fi.append(output[0].detach().float())
            fc.append(output[1].detach().float())
            ii.append(torch.as_tensor(image_ids).reshape(-1).cpu())
            ai.append(torch.as_tensor(ann_ids).reshape(-1).cpu())
        ex = collect(torch.cat(fi), torch.cat(fc), torch.cat(ii), torch.cat(ai), num_images, num_captions, iid_to_cls)
        if iid_to_cls:
            print(f"Num images ({num_images}) -> Num classes ({len(set(ex['image_classes'].tolist()))})")
        return ex

    @torch.no_grad()
    def retrieve(self, q_features, g_features, q_ids, g_ids, q_classes=None, g_classes=None, topk=10, batch_size=1024):
        """eval_coco.py:250-294 (top-k gallery ids and negated similarities per query; not on the evaluation path)"""
        if len(q_features) != len(q_ids):
            raise RuntimeError("length mismatch {}, {}".format(q_features.shape, np.shape(q_ids)))
        if len(g_features) != len(g_ids):
            raise RuntimeError("length mismatch {}, {}".format(g_features.shape, np.shape(g_ids)))
        q_ids, g_ids = np.array(q_ids), np.array(g_ids)
        dev = torch.device(self.eval_device)
        g = torch.as_tensor(g_features).to(dev, torch.float64).reshape(len(g_ids), -1)
        q = torch.as_tensor(q_features).to(dev, torch.float64).reshape(len(q_ids), -1)
        retrieved_items, retrieved_scores = {}, {}
        if dev.type != "cuda":
            raise _lib.FedcolaHipError("fedcola_amd retrieval runs on the GPU only (eval_device must be a cuda device)")
        q, g = q.contiguous(), g.contiguous()
        L = _lib.lib()
        for b0 in range(0, len(q_ids), batch_size):
            qb = q[b0:b0 + batch_size]
            sim = torch.empty(qb.shape[0], g.shape[0], dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):        # similarities by the library's fp64 MFMA kernel (the one evaluate_recall ranks with); torch only orders them
                _lib.check(L.fc_k_sim_f64(_lib.ptr(qb), _lib.ptr(g), _lib.ptr(sim), qb.shape[0], g.shape[0], qb.shape[1], _lib.stream_ptr()))
            sims, pred = sim.neg_().sort(stable=True)
            for r in range(sims.shape[0]):
                retrieved_items[q_ids[b0 + r]] = [item for item in g_ids[pred[r, :topk].cpu().numpy()]]
                retrieved_scores[q_ids[b0 + r]] = sims[r][:topk].cpu().numpy()
        return retrieved_items, retrieved_scores, None

    @torch.no_grad()
    def evaluate_recall(self, q_features, g_features, q_labels, g_labels, q_ids=None, g_ids=None, batch_size=1024):
        """eval_coco.py:296-351"""
        if len(q_features) != len(q_labels):
            raise RuntimeError("length mismatch {}, {}".format(q_features.shape, q_labels.shape))
        if len(g_features) != len(g_labels):
            raise RuntimeError("length mismatch {}, {}".format(g_features.shape, g_labels.shape))
        best_pred_ranks = best_ranks_device(q_features, g_features, q_labels, g_labels, self.eval_device, batch_size)
        recall_1 = recall_at_k(best_pred_ranks, 1)
        recall_5 = recall_at_k(best_pred_ranks, 5)
        recall_10 = recall_at_k(best_pred_ranks, 10)
        medr = np.floor(np.median(best_pred_ranks)) + 1
        meanr = np.mean(best_pred_ranks) + 1
        return {"recall_1": recall_1, "recall_5": recall_5, "recall_10": recall_10, "rsum": recall_1 + recall_5 + recall_10,
                "medr": medr, "meanr": meanr}

    def evaluate_n_fold(self, extracted_features, n_crossfolds, n_images_per_crossfold, n_captions_per_crossfold, eval_batch_size):
        """eval_coco.py:353-407"""
        image_features = extracted_features["image_features"]
        caption_features = extracted_features["caption_features"]
        image_classes = extracted_features["image_classes"]
        caption_classes = extracted_features["caption_classes"]
        keys = ("recall_1", "recall_5", "recall_10", "rsum", "medr", "meanr")
        n_fold_scores = {task: {k: [] for k in keys} for task in ("i2t", "t2i")}
        for idx in range(n_crossfolds):
            if self.logger:
                self.logger.info("evaluating {}-th fold".format(idx + 1))
            isl = np.arange(idx * n_images_per_crossfold, (idx + 1) * n_images_per_crossfold)
            csl = np.arange(idx * n_captions_per_crossfold, (idx + 1) * n_captions_per_crossfold)
            fi, ci = image_features[isl], image_classes[isl]
            fc, cc = caption_features[csl], caption_classes[csl]
            _scores = {"i2t": self.evaluate_recall(fi, fc, ci, cc, batch_size=eval_batch_size),
                       "t2i": self.evaluate_recall(fc, fi, cc, ci, batch_size=eval_batch_size)}
            for _task, _task_scores in _scores.items():
                for key, val in _task_scores.items():
                    n_fold_scores[_task][key].append(val)
        return {_task: {key: np.mean(np.array(val)) for key, val in _task_scores.items()} for _task, _task_scores in n_fold_scores.items()}

    @torch.no_grad()
    def evaluate(self, dataloader, n_crossfolds=None, n_images_per_crossfold=1000, n_captions_per_crossfold=5000, eval_batch_size=1024,
                 key=None):
        """eval_coco.py:409-469"""
        scores = {}
        if self.logger:
            self.logger.info("extracting features...")
        extracted_features = self.extract_features(dataloader)
        image_features = extracted_features["image_features"]
        caption_features = extracted_features["caption_features"]
        image_classes = extracted_features["image_classes"]
        caption_classes = extracted_features["caption_classes"]
        scores["mean_log_image_sigma"] = np.mean(extracted_features["image_sigmas"])
        scores["mean_log_caption_sigma"] = np.mean(extracted_features["caption_sigmas"])
        if n_crossfolds is None:
            n_crossfolds = self.n_crossfolds
        if dataloader.dataset.iid_to_cls:
            print('"use_class" setting does not evaluate 1k crossfolds')
            n_crossfolds = -1
        if n_crossfolds > 0:
            scores["n_fold"] = self.evaluate_n_fold(extracted_features, n_crossfolds, n_images_per_crossfold, n_captions_per_crossfold,
                                                    eval_batch_size)
        if self.logger:
            self.logger.info("evaluating i2t...")
        scores["i2t"] = self.evaluate_recall(image_features, caption_features, image_classes, caption_classes, batch_size=eval_batch_size)
        if self.logger:
            self.logger.info("evaluating t2i...")
        scores["t2i"] = self.evaluate_recall(caption_features, image_features, caption_classes, image_classes, batch_size=eval_batch_size)
        for key in ("rsum", "medr", "meanr"):
            scores[key] = scores["i2t"][key] + scores["t2i"][key]
        return scores
