#!/usr/bin/env python
"""First-batch latency of a client round (development aid): time from iter(PinnedBatchLoader) to the first assembled batch, per worker
count, and the H2D copy of that batch.  usage: tools/first_batch_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fedcola_amd.loaders.batch import PinnedBatchLoader
B, seq = 64, 32
ds = bench.InMemoryPairs(20 * B, seq, 7732)
out = {}
for w in (4, 8, 16, 32, 64):
    ld = PinnedBatchLoader(ds, B, shuffle=True, workers=w)
    firsts, nexts = [], []
    for rep in range(5):
        t0 = time.perf_counter(); it = iter(ld); b = next(it); t1 = time.perf_counter(); b2 = next(it); b3 = next(it); t2 = time.perf_counter()
        firsts.append((t1 - t0) * 1e3); nexts.append((t2 - t1) * 1e3 / 2)
        del it
    out[f"workers={w}"] = dict(first_batch_ms=round(min(firsts[1:]), 2), next_batch_ms=round(min(nexts[1:]), 2))
img = b[0]
d = torch.empty_like(img, device="cuda")
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); d.copy_(img, non_blocking=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
out["h2d_38MB_ms"] = round(min(ts), 2)
# raw gather speed of one thread and of torch's own parallel index_select
i = torch.randperm(20 * B)[:B]
o = torch.empty((B,) + tuple(ds.img.shape[1:]), pin_memory=True)
for nt in (1, 8, 32):
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    for _ in range(5): torch.index_select(ds.img, 0, i, out=o)
    out[f"index_select_{nt}_threads_ms"] = round((time.perf_counter() - t0) * 1e3 / 5, 2)
print(json.dumps(out))
