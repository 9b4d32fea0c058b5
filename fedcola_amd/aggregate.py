"""Server-side aggregation (/root/reference/src/server/fedavgserver.py:591-668) as a device blend + RCCL all-reduce.

The reference blends sequentially on the CPU: for every sampled client in ascending id order and every key,
``g <- g + (theta_i - g) * c_i`` (skipping clients that lack the key or have c_i == 0).  That recurrence has the closed form

    g_new = w_g * g + sum_j w_j * theta_j,     w_j = c_j * prod_{l > j} (1 - c_l),     w_g = prod_l (1 - c_l)

over the participating clients j (in blend order).  The host computes the coefficient table exactly like the reference
(scope table, compensation denominators, out-modality scaling -- quirks included) and the effective weights; each rank
then forms its local partial  [rank 0: w_g*g] + sum_{local clients} w_j*theta_j  with one HIP kernel over the flat
parameter buffers (``fc_aggregate_blend``), and the partials are summed with ONE all-reduce over xGMI.  Every rank ends
with the new global model in HBM, so next round's ``download`` is a device-to-device copy.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Callable, Dict, List, Mapping, Optional, Sequence

import torch

# ---------------------------------------------------------------------------------------------- scope bookkeeping


def get_name_type(name: str) -> str:
    """fedavgserver.py:94-104 ('mlp' is unreachable for 'blockses.*' keys: 'blocks' matches first)."""
    if "embeddings" in name:
        return "embedding"
    elif "attention" in name or "attn" in name:
        return "attn"
    elif "blocks" in name:
        return "blocks"
    elif "mlp" in name:
        return "mlp"
    return "task"


def get_first_number(string):
    m = re.search(r"\d+", string)
    return int(m.group()) if m else None


def get_name_modality(name, modalities):
    """fedavgserver.py:113-115."""
    idx = get_first_number(name)
    return modalities[idx] if idx is not None else None


def init_param_scope(param_names: Sequence[str], shared_param: str, share_scope: str) -> Dict[str, str]:
    """fedavgserver.py:183-238."""
    scope: Dict[str, str] = {}
    if shared_param not in ("none", "attn", "blocks", "mlp"):
        return scope
    for name in param_names:
        t = get_name_type(name)
        scope[name] = share_scope if (shared_param != "none" and t == shared_param) else "dataset"
    return scope


def mixing_coefficients(keys, param_scope, updated_sizes: Mapping[int, int], clients, *, dataset, task, modality,
                        out_modality_scale=1, args=None, fedavg=False) -> Dict[str, Dict[int, float]]:
    """The coefficient table of fedavgserver.py:601-653 (both branches), quirks preserved:
    'modality' scope is a substring-overlap test (:631); with --compensation and share_scope == 'modality' the denominator is
    the total size of modality-overlapping clients for EVERY key (:643-645); without compensation the denominator counts
    clients that pass the scope test even if they lack the key (:653)."""
    coefficients: Dict[str, Dict[int, float]] = {}
    for param_name in keys:
        new_num: Dict[int, float] = {}
        old_sum = sum(updated_sizes.values())
        sc = param_scope[param_name]
        param_modality = None if fedavg else get_name_modality(param_name, args.modalities)
        for identifier, num in updated_sizes.items():
            c = clients[identifier]
            if sc == "all":
                new_num[identifier] = num
            elif sc == "dataset":
                new_num[identifier] = num if c.dataset == dataset else 0
            elif sc == "task":
                new_num[identifier] = num if c.task == task else 0
            elif sc == "modality":
                if fedavg:
                    new_num[identifier] = num if c.modality == modality else 0
                else:
                    new_num[identifier] = num if (c.modality in modality or modality in c.modality) else 0
            elif sc == "modality_exact" and not fedavg:
                new_num[identifier] = num if (c.modality == param_modality or param_modality in c.modality) else 0
            if (not fedavg) and c.modality != modality and out_modality_scale != 1:
                old_sum -= new_num[identifier]
                new_num[identifier] *= out_modality_scale
                old_sum += new_num[identifier]
        if (not fedavg) and args.compensation:
            if args.share_scope == "all":
                coefficients[param_name] = {i: float(n / old_sum) for i, n in new_num.items()}
            elif args.share_scope == "modality":
                comp = sum(s for i, s in updated_sizes.items() if clients[i].modality in modality or modality in clients[i].modality)
                coefficients[param_name] = {i: float(n / comp) if comp != 0 else 0 for i, n in new_num.items()}
            elif args.share_scope == "modality_exact":
                last = list(updated_sizes.keys())[-1]               # the reference reads the leaked loop variable here (:648)
                if param_modality:
                    comp = sum(s for i, s in updated_sizes.items()
                               if clients[i].modality == param_modality or param_modality in clients[last].modality)
                else:
                    comp = sum(s for i, s in updated_sizes.items() if clients[i].modality in modality or modality in clients[i].modality)
                coefficients[param_name] = {i: float(n / comp) if comp != 0 else 0 for i, n in new_num.items()}
        else:
            tot = sum(new_num.values())
            coefficients[param_name] = {i: float(n / tot) if tot != 0 else 0 for i, n in new_num.items()}
    return coefficients


def effective_weights(cs: Sequence[float]):
    """Closed form of the sequential blend: returns (w_g, [w_j]) for the participating coefficients in blend order."""
    w = [0.0] * len(cs)
    tail = 1.0
    for j in reversed(range(len(cs))):
        w[j] = cs[j] * tail
        tail *= (1.0 - cs[j])
    return tail, w


# ---------------------------------------------------------------------------------------------- blend plan


@dataclass
class BlendPlan:
    keys: List[str]
    ids: List[int]                   # sampled client ids (blend order)
    seg_off: torch.Tensor            # int64 [nseg]   offsets in the GLOBAL model's flat buffer
    seg_len: torch.Tensor            # int64 [nseg]
    src_off: torch.Tensor            # int64 [nseg, m] offsets in each client's own flat buffer (-1: key absent / not participating)
    weights: torch.Tensor            # float32 [nseg, m+1]  (w_g, w_1..w_m)


def build_plan(global_model, ids: Sequence[int], coefficients, client_segments: Mapping[int, Mapping[str, dict]],
               zero_init: bool = False) -> BlendPlan:
    """client_segments[i][key] -> {'offset':..} of the keys client i uploads (aux / scale keys already dropped).
    zero_init: CreamflServer._aggregate (creamflserver.py:257-288) -- a plain weighted sum into zeros: w_g = 0, w_j = c_j."""
    keys = list(coefficients.keys())
    m = len(ids)
    seg_off = torch.empty(len(keys), dtype=torch.int64)
    seg_len = torch.empty(len(keys), dtype=torch.int64)
    src_off = torch.full((len(keys), m), -1, dtype=torch.int64)
    weights = torch.zeros(len(keys), m + 1, dtype=torch.float32)
    gseg = global_model.segments
    for s, k in enumerate(keys):
        seg_off[s] = gseg[k]["offset"]
        seg_len[s] = gseg[k]["numel"]
        part = [(j, i) for j, i in enumerate(ids) if k in client_segments[i] and coefficients[k][i] != 0]
        wg, w = (0.0, [coefficients[k][i] for _, i in part]) if zero_init else effective_weights([coefficients[k][i] for _, i in part])
        weights[s, 0] = wg
        for (j, i), wj in zip(part, w):
            weights[s, 1 + j] = wj
            src_off[s, j] = client_segments[i][k]["offset"]
    return BlendPlan(keys, list(ids), seg_off, seg_len, src_off, weights)


def hip_local_partial(plan: BlendPlan, global_flat: torch.Tensor, local_flats: Mapping[int, torch.Tensor], include_global: bool) -> torch.Tensor:
    """This rank's share of the blend with one HIP kernel: [w_g*g if include_global] + sum_{local j} w_j*theta_j, laid out like
    the global flat buffer (zeros outside the plan's segments)."""
    from . import _lib
    from ._lib import check, ptr
    dev = global_flat.device
    m = len(plan.ids)
    w = plan.weights.clone()
    src = plan.src_off.clone()
    bases = torch.zeros(m, dtype=torch.int64)
    for j, i in enumerate(plan.ids):
        if i in local_flats:
            bases[j] = local_flats[i].data_ptr()
        else:
            w[:, 1 + j] = 0.0
            src[:, j] = -1
    if not include_global:
        w[:, 0] = 0.0
    out = torch.zeros_like(global_flat)
    d = [t.to(dev) for t in (bases, plan.seg_off, plan.seg_len, src.contiguous(), w.contiguous())]
    check(_lib.lib().fc_aggregate_blend(ptr(out), ptr(global_flat), ptr(d[0]), m, ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]),
                                        len(plan.keys), _lib.stream_ptr()))
    torch.cuda.current_stream().synchronize()          # the tables above are temporaries
    return out


def _copy_runs(plan: BlendPlan):
    """Planned segments merged into maximal runs: neighbours separated only by alignment padding (< 64 elements, so no unplanned
    segment can sit in between) are copied together -- a handful of device copies instead of one per state_dict key."""
    runs = getattr(plan, "_runs", None)
    if runs is None:
        segs = sorted((int(o), int(o) + int(n)) for o, n in zip(plan.seg_off.tolist(), plan.seg_len.tolist()))
        runs = []
        for o, e in segs:
            if runs and 0 <= o - runs[-1][1] < 64:
                runs[-1][1] = e
            else:
                runs.append([o, e])
        plan._runs = runs
    return runs


def aggregate(global_model, plan: BlendPlan, local_flats: Mapping[int, torch.Tensor], *, rank: int = 0, world: int = 1,
              all_reduce: Optional[Callable[[torch.Tensor], None]] = None, local_partial=hip_local_partial):
    """Blend into ``global_model`` in place.  With world > 1 every rank contributes the clients it trained and the
    partials are summed by ``all_reduce`` (torch.distributed.all_reduce over RCCL by default)."""
    g = global_model.flat.data
    partial = local_partial(plan, g, local_flats, include_global=(rank == 0))
    if world > 1:
        if all_reduce is None:
            import torch.distributed as dist
            all_reduce = dist.all_reduce
        all_reduce(partial)
    for o, e in _copy_runs(plan):                        # only the planned (required_params) segments are replaced
        g[o:e].copy_(partial[o:e])
    global_model._bump()
    return global_model
