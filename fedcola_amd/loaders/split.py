"""``simulate_split`` (/root/reference/src/loaders/split.py:10-165): sample indices -> clients.

Index bookkeeping must be bit-exact: every branch draws from ``np.random`` in the reference's order (permutation, then the
keep ratios, ...), so under the same ``np.random.seed`` the maps are identical (tests/golden/split.json).
  iid         :20-30   permutation, array_split
  unbalanced  :33-74   permutation, array_split, keep ratio U[0.95, 0.99) per client; for Flickr30k / Coco the permutation is
                       over IMAGES (len // 5) and every kept image expands to its 5 caption indices (5i .. 5i+4)
  patho       :77-130  class shards (McMahan et al.), `mincls` classes per client
  diri        :132-161 per-class Dirichlet(cncntrtn) proportions with the balance mask, retried until every client has >= 10
"""
from __future__ import annotations

import logging

import numpy as np

logger = logging.getLogger(__name__)


def _keep(split_indices):
    keep_ratio = np.random.uniform(low=0.95, high=0.99, size=len(split_indices))
    return [indices[:int(len(indices) * ratio)] for indices, ratio in zip(split_indices, keep_ratio)]


def simulate_split(args, dataset):
    K = args.K
    if args.split_type == "iid":
        parts = np.array_split(np.random.permutation(len(dataset)), K)
        return {k: parts[k] for k in range(K)}

    caption_sets = args.dataset in ["Flickr30k", "Coco"]
    if args.split_type == "unbalanced" or (caption_sets and args.split_type != "iid"):
        if caption_sets:
            parts = _keep(np.array_split(np.random.permutation(len(dataset) // 5), K))
            # image index i -> its five caption samples, in order (plain Python ints like the reference's list of appends)
            parts = [(np.asarray(p, dtype=np.int64)[:, None] * 5 + np.arange(5)).reshape(-1).tolist() for p in parts]
        else:
            parts = _keep(np.array_split(np.random.permutation(len(dataset)), K))
        return {k: parts[k] for k in range(K)}

    if args.split_type == "patho":
        if args.mincls < 2:
            logger.error("[SIMULATE] Each client should have samples from at least 2 distinct classes!")
            raise AssertionError
        _, inverse, counts = np.unique(dataset.targets, return_inverse=True, return_counts=True)
        class_indices = np.split(np.argsort(inverse), np.cumsum(counts[:-1]))
        shards_per_class = K * args.mincls // args.num_classes
        if shards_per_class < 1:
            raise Exception(f"[SIMULATE] Increase the number of minimum class (`args.mincls` > {args.mincls}) or the number of "
                            f"participating clients (`args.K` > {args.K})!")
        shards = [np.array_split(np.random.permutation(idx), shards_per_class) for idx in class_indices]
        remaining = dict(zip(range(args.num_classes), [len(s) for s in shards]))
        assigned = []
        for _ in range(K):
            prob = np.where(np.array(list(remaining.values())) > 0, 1., 0.)
            prob /= sum(prob)
            try:
                chosen = np.random.choice(args.num_classes, args.mincls, replace=False, p=prob)
            except Exception:        # fewer classes with shards left than mincls
                chosen = np.random.choice(args.num_classes, args.mincls, replace=True, p=prob)
            mine = []
            for cls in chosen:
                pick = np.random.choice(len(shards[cls]), 1)[0]
                mine.append(shards[cls].pop(pick))
                remaining[cls] -= 1
            assigned.append(np.concatenate(mine))
        return {k: assigned[k] for k in range(K)}

    if args.split_type == "diri":
        y = np.array(dataset.targets)
        N, n_nets = len(dataset.targets), K
        min_size = 0
        while min_size < 10:
            batches = [[] for _ in range(n_nets)]
            for cls in range(args.num_classes):
                idx = np.where(y == cls)[0]
                np.random.shuffle(idx)
                prop = np.random.dirichlet(np.repeat(args.cncntrtn, n_nets))
                prop = np.array([p * (len(b) < N / n_nets) for p, b in zip(prop, batches)])
                prop = prop / prop.sum()
                cuts = (np.cumsum(prop) * len(idx)).astype(int)[:-1]
                batches = [b + part.tolist() for b, part in zip(batches, np.split(idx, cuts))]
                min_size = min(len(b) for b in batches)
        out = {}
        for j in range(n_nets):
            np.random.shuffle(batches[j])
            out[j] = batches[j]
        return out

    if args.split_type == "leaf":
        logger.info("[SIMULATE] Use pre-defined split!")
        return None
    raise NotImplementedError(args.split_type)
