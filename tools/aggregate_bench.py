#!/usr/bin/env python
"""Aggregation kernels on one GPU: m local clients x the ViT-S img+txt flat buffer (184 MB) blended into the global model.
closed form (fc_aggregate, k_blend_v): reads (m + 1) buffers, writes 1;  exact order (fc_aggregate_blend_seq, k_blend_seq): the same streams,
sequential rounding.  Effective GB/s = (m + 2) x 184 MB / time.   usage: tools/aggregate_bench.py [m=8] [reps=10]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import Args
from fedcola_amd import aggregate as agg
from fedcola_amd.mome import create_model
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
a = Args(); a.precision = "bf16"
dev = torch.device("cuda")
g = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
n = g.flat.numel()
clients = {i: torch.randn(n, device=dev) * 0.02 for i in range(m)}
ids = list(range(m))
keys = list(g.required_params().keys())
coef = {k: {i: 1.0 / m for i in ids} for k in keys}
segs = {i: g.segments for i in ids}
plan = agg.build_plan(g, ids, coef, segs)
out = {}


def timeit(f):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timeit(lambda: agg.aggregate(g, plan, clients, rank=0, world=1))
out["closed_form"] = dict(ms=round(t * 1e3, 3), effective_GBps=round((m + 2) * 4 * n / t / 1e9, 1))
t = timeit(lambda: agg.aggregate_exact(g, keys, ids, coef, segs, clients))
out["exact_order"] = dict(ms=round(t * 1e3, 3), effective_GBps=round((m + 2) * 4 * n / t / 1e9, 1),
                          note="includes the per-call table upload and stream synchronisation of the verification path")
print(json.dumps(dict(clients=m, buffer_MB=round(4 * n / 1e6, 1), segments=len(plan.keys), **out)))
