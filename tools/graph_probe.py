#!/usr/bin/env python
"""Experiment: capture one fc_client_step into a HIP graph (torch.cuda.CUDAGraph) and time replays against eager launches.
The captured AdamW uses a frozen step number: timing only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import Args, make_batch
from fedcola_amd import _lib
from fedcola_amd.mome import create_model
a = Args(); a.precision = "bf16"
dev = torch.device("cuda")
torch.manual_seed(1)
model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
model.train()
B, seq = 64, a.seq_len
img, ids = make_batch(B, seq, a.vocab_size, 0, dev)
n = model.flat.numel()
grads = torch.zeros(n, device=dev); m1 = torch.zeros(n, device=dev); m2 = torch.zeros(n, device=dev)
lossbuf = torch.zeros(2, device=dev)
model.prepare_weights(force=True)
ws = model.workspace(B, seq)
L, P = _lib.lib(), _lib.ptr
def step(k, sp):
    _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), None,
                                B, seq, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lossbuf), P(ws), ws.numel(), sp))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    sp = _lib.stream_ptr()
    for k in range(1, 6):
        step(k, sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(6, 36):
        step(k, sp)
    torch.cuda.synchronize()
    print("eager  ms/step %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            step(36, _lib.stream_ptr())
    except Exception as e:
        print("capture failed:", repr(e)[:500]); sys.exit(0)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("graph  ms/step %.3f (host enqueue %.3f)" % ((time.perf_counter() - t0) / 30 * 1e3, (t1 - t0) / 30 * 1e3), "loss", float(lossbuf[1]))
