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

import ctypes as C
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


def _alias_map(model) -> Dict[str, str]:
    """alias key -> key of the segment it views (sync_shared_weights: scope 'all' / colearn_param 'attn'); {} for plain models."""
    f = getattr(model, "_alias_keys", None)
    return dict(f()) if f is not None else {}


def build_plan(global_model, ids: Sequence[int], coefficients, client_segments: Mapping[int, Mapping[str, dict]],
               zero_init: bool = False) -> BlendPlan:
    """client_segments[i][key] -> {'offset':..} of the keys client i uploads (aux / scale keys already dropped; alias keys of a
    colearn_param == 'attn' model point at the owner's segment).
    zero_init: CreamflServer._aggregate (creamflserver.py:257-288) -- a plain weighted sum into zeros: w_g = 0, w_j = c_j.

    Alias keys: in the reference the two keys of a shared tensor ARE one tensor in ``final_sd`` (required_params() is a shallow copy
    of state_dict(), mome.py:844-860), so the in-place loop (fedavgserver.py:656-664: clients outer, keys inner) blends it once per
    key for every client: client 1 under the owner's key, client 1 again under the alias key, client 2 under the owner's key, ...
    each step with that key's coefficient.  One plan row per tensor: the closed form over that interleaved sequence."""
    alias = _alias_map(global_model)
    gseg = global_model.segments
    groups: "Dict[str, List[str]]" = {}
    for k in coefficients.keys():
        groups.setdefault(alias.get(k, k), []).append(k)
    keys = list(groups.keys())
    m = len(ids)
    seg_off = torch.empty(len(keys), dtype=torch.int64)
    seg_len = torch.empty(len(keys), dtype=torch.int64)
    src_off = torch.full((len(keys), m), -1, dtype=torch.int64)
    weights = torch.zeros(len(keys), m + 1, dtype=torch.float32)
    for s, owner in enumerate(keys):
        seg_off[s] = gseg[owner]["offset"]
        seg_len[s] = gseg[owner]["numel"]
        if zero_init and len(groups[owner]) > 1:
            raise NotImplementedError("zero-initialised aggregation of a model with shared (alias) tensors")
        seq = [(j, i, k) for j, i in enumerate(ids) for k in groups[owner] if k in client_segments[i] and coefficients[k][i] != 0]
        cs = [coefficients[k][i] for _, i, k in seq]
        wg, w = (0.0, cs) if zero_init else effective_weights(cs)
        weights[s, 0] = wg
        for (j, i, k), wj in zip(seq, w):
            weights[s, 1 + j] += wj
            off = int(client_segments[i][k]["offset"])
            assert src_off[s, j] in (-1, off), "a client's keys of one shared tensor must point at one segment"
            src_off[s, j] = off
    return BlendPlan(keys, list(ids), seg_off, seg_len, src_off, weights)


def _device_tables(plan: BlendPlan, dev, local_ids, include_global: bool):
    """The plan's tables on `dev`, masked for the clients this rank holds: built once per (device, local set) and kept on the plan
    (the plan of a round is reused for every global model's all-reduce; bench.py reuses one plan for all rounds)."""
    cache = plan.__dict__.setdefault("_dev", {})
    key = (str(dev), tuple(sorted(local_ids)), bool(include_global))
    t = cache.get(key)
    if t is None:
        w = plan.weights.clone()
        src = plan.src_off.clone()
        for j, i in enumerate(plan.ids):
            if i not in local_ids:
                w[:, 1 + j] = 0.0
                src[:, j] = -1
        if not include_global:
            w[:, 0] = 0.0
        runs = _copy_runs(plan)
        t = dict(seg_off=plan.seg_off.to(dev), seg_len=plan.seg_len.to(dev), src=src.contiguous().to(dev), w=w.contiguous().to(dev),
                 run_off=(C.c_int64 * len(runs))(*[o for o, _ in runs]), run_len=(C.c_int64 * len(runs))(*[e - o for o, e in runs]),
                 n_runs=len(runs))
        cache[key] = t
    return t


def _bases_array(plan: BlendPlan, local_flats):
    m = len(plan.ids)
    arr = (C.c_void_p * max(m, 1))()
    for j, i in enumerate(plan.ids):
        arr[j] = local_flats[i].data_ptr() if i in local_flats else None
    return arr


def _partial_buffer(global_model):
    """numel-float scratch kept on the model (zero once: ranges outside the plan stay zero through every all-reduce)."""
    g = global_model.flat.data
    p = getattr(global_model, "_agg_partial", None)
    if p is None or p.device != g.device or p.numel() != g.numel():
        p = torch.zeros_like(g)
        global_model._agg_partial = p
    return p


def hip_local_partial(plan: BlendPlan, global_flat: torch.Tensor, local_flats: Mapping[int, torch.Tensor], include_global: bool,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """This rank's share of the blend with one HIP kernel: [w_g*g if include_global] + sum_{local j} w_j*theta_j, laid out like
    the global flat buffer (zeros outside the plan's segments).  No host synchronisation: the tables are cached on the plan and
    the client pointers travel as kernel arguments."""
    from . import _lib
    from ._lib import check, ptr
    if len(plan.ids) > 64:
        return _hip_local_partial_many(plan, global_flat, local_flats, include_global, out)
    t = _device_tables(plan, global_flat.device, set(local_flats.keys()), include_global)
    if out is None:
        out = torch.zeros_like(global_flat)
    check(_lib.lib().fc_aggregate_partial(ptr(out), ptr(global_flat), _bases_array(plan, local_flats), len(plan.ids), ptr(t["seg_off"]),
                                          ptr(t["seg_len"]), ptr(t["src"]), ptr(t["w"]), len(plan.keys), _lib.stream_ptr()))
    return out


def _hip_local_partial_many(plan, global_flat, local_flats, include_global, out=None):
    """More than 64 sampled clients: the client pointers go through a device table (fc_aggregate_blend).  Writes `out` when given
    (aggregate_many hands in its slice of the concatenated all-reduce buffer)."""
    from . import _lib
    from ._lib import check, ptr
    dev = global_flat.device
    t = _device_tables(plan, dev, set(local_flats.keys()), include_global)
    bases = torch.tensor([local_flats[i].data_ptr() if i in local_flats else 0 for i in plan.ids], dtype=torch.int64).to(dev)
    if out is None:
        out = torch.zeros_like(global_flat)
    check(_lib.lib().fc_aggregate_blend(ptr(out), ptr(global_flat), ptr(bases), len(plan.ids), ptr(t["seg_off"]), ptr(t["seg_len"]), ptr(t["src"]),
                                        ptr(t["w"]), len(plan.keys), _lib.stream_ptr()))
    torch.cuda.current_stream().synchronize()          # `bases` is a temporary
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
              all_reduce: Optional[Callable[[torch.Tensor], None]] = None, local_partial=None, comm=None):
    """Blend into ``global_model`` in place.  With world > 1 every rank contributes the clients it trained and the partials are
    summed over the ranks: through the C ABI's own RCCL communicator when ``comm`` (fedcola_amd.comm.Comm) is given
    (``fc_aggregate``: blend + ncclAllReduce + copy-back in one call), else by ``all_reduce`` (torch.distributed.all_reduce over RCCL
    by default).  ``local_partial`` replaces the HIP blend (CPU tests)."""
    g = global_model.flat.data
    if local_partial is None and g.is_cuda and len(plan.ids) <= 64 and (world == 1 or comm is not None):
        from . import _lib
        from ._lib import check, ptr
        t = _device_tables(plan, g.device, set(local_flats.keys()), rank == 0)
        partial = _partial_buffer(global_model) if world > 1 else g
        check(_lib.lib().fc_aggregate(comm.h if (comm is not None and world > 1) else None, ptr(g), ptr(partial), g.numel(),
                                      _bases_array(plan, local_flats), len(plan.ids), ptr(t["seg_off"]), ptr(t["seg_len"]), ptr(t["src"]),
                                      ptr(t["w"]), len(plan.keys), t["run_off"], t["run_len"], t["n_runs"], _lib.stream_ptr()))
        global_model._bump()
        return global_model
    if local_partial is None:
        partial = hip_local_partial(plan, g, local_flats, include_global=(rank == 0), out=_partial_buffer(global_model) if world > 1 else None)
    else:
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


def aggregate_many(items, *, rank: int = 0, world: int = 1, all_reduce: Optional[Callable[[torch.Tensor], None]] = None, comm=None,
                   local_partial=None):
    """Several global models of one round (one per dataset, fedavgserver.py:812-819) with ONE cross-rank sum: every model's local
    partial is blended into its range of a concatenated buffer, the buffer is all-reduced once (K models: one collective instead of K
    -- on xGMI a ring all-reduce is latency- and per-link-bound, fewer and larger is the cheaper shape), and each model takes its
    planned ranges back.  items: [(global_model, plan, local_flats)].  ``local_partial`` replaces the HIP blend (CPU tests)."""
    if world == 1 or len(items) <= 1:
        kw = {} if local_partial is None else {"local_partial": local_partial}
        for gm, plan, flats in items:
            aggregate(gm, plan, flats, rank=rank, world=world, all_reduce=all_reduce, comm=comm, **kw)
        return
    g0 = items[0][0].flat.data
    total = sum(gm.flat.numel() for gm, _, _ in items)
    owner = items[0][0]
    cat = getattr(owner, "_agg_cat", None)
    if cat is None or cat.device != g0.device or cat.numel() != total:
        cat = torch.zeros(total, device=g0.device)      # zero once: ranges outside the plans stay zero through every all-reduce
        owner._agg_cat = cat
    off = 0
    views = []
    for gm, plan, flats in items:
        n = gm.flat.numel()
        if local_partial is None:
            part = hip_local_partial(plan, gm.flat.data, flats, include_global=(rank == 0), out=cat[off:off + n])
            if part.data_ptr() != cat[off:off + n].data_ptr():      # a path that ignored `out` must not leave the slice stale
                cat[off:off + n].copy_(part)
        else:
            cat[off:off + n].copy_(local_partial(plan, gm.flat.data, flats, include_global=(rank == 0)))
        views.append((gm, plan, cat[off:off + n]))
        off += n
    if comm is not None:
        comm.all_reduce(cat)
    else:
        if all_reduce is None:
            import torch.distributed as dist
            all_reduce = dist.all_reduce
        all_reduce(cat)
    for gm, plan, part in views:
        g = gm.flat.data
        for o, e in _copy_runs(plan):
            g[o:e].copy_(part[o:e])
        gm._bump()


def aggregate_exact(global_model, plan_keys: Sequence[str], ids: Sequence[int], coefficients, client_segments, local_flats, *, comm=None):
    """The reference's sequential in-place blend itself, in its order and rounding (``fc_aggregate_blend_seq`` /
    ``fc_aggregate_exact``): bit-identical to fedavgserver.py:656-664 in fp32.  Verification mode of the closed form.
    Single process: every sampled client is local.  With ``comm``: one client per rank, ids[r] trained on rank r; the client buffers
    are all-gathered (padded to the longest) and blended in rank (= ascending id) order on every rank."""
    from . import _lib
    from ._lib import check, ptr
    g = global_model.flat.data
    dev = g.device
    m = len(ids)
    gseg = global_model.segments
    alias = _alias_map(global_model)
    # a shared tensor listed under two keys is blended once per key for every client, clients outer (the reference's loop): each client
    # becomes `nrep` consecutive virtual clients, one per occurrence of the tensor among the keys
    groups: "Dict[str, List[str]]" = {}
    for k in plan_keys:
        groups.setdefault(alias.get(k, k), []).append(k)
    owners = list(groups.keys())
    nrep = max(len(v) for v in groups.values()) if groups else 1
    L = _lib.lib()
    seg_off = torch.tensor([gseg[o]["offset"] for o in owners], dtype=torch.int64)
    seg_len = torch.tensor([gseg[o]["numel"] for o in owners], dtype=torch.int64)
    src = torch.full((len(owners), m * nrep), -1, dtype=torch.int64)
    coef = torch.zeros(len(owners), m * nrep, dtype=torch.float32)
    for s, o in enumerate(owners):
        for j, i in enumerate(ids):
            for r, k in enumerate(groups[o]):
                if k in client_segments[i] and coefficients[k][i] != 0:
                    src[s, j * nrep + r] = client_segments[i][k]["offset"]
                    coef[s, j * nrep + r] = coefficients[k][i]
    d = [t.to(dev) for t in (seg_off, seg_len, src.contiguous(), coef.contiguous())]
    if comm is None or comm.world == 1:
        assert all(i in local_flats for i in ids), "single-process exact blend needs every sampled client's weights"
        assert m * nrep <= 64, "exact blend: too many (virtual) clients for one launch"
        arr = (C.c_void_p * max(m * nrep, 1))(*[local_flats[i].data_ptr() for i in ids for _ in range(nrep)])
        check(L.fc_aggregate_blend_seq(ptr(g), arr, m * nrep, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), len(owners), _lib.stream_ptr()))
    else:
        assert m == comm.world and len(local_flats) == 1, "exact mode across ranks: one client per rank"
        assert nrep == 1, "exact mode across ranks: models with shared (alias) tensors are blended in a single process"
        (mine, flat), = local_flats.items()
        assert ids[comm.rank] == mine
        slot = max(max(int(sg["offset"]) + int(sg["numel"]) for sg in client_segments[i].values()) for i in ids)
        local = torch.zeros(slot, device=dev)
        local[: min(slot, flat.numel())].copy_(flat[: min(slot, flat.numel())])
        gathered = torch.empty(slot * comm.world, device=dev)
        check(L.fc_aggregate_exact(comm.h, ptr(g), ptr(local), ptr(gathered), slot, ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), len(owners),
                                   _lib.stream_ptr()))
    torch.cuda.current_stream().synchronize()            # the tables above are temporaries of this (verification) call
    global_model._bump()
    return global_model
