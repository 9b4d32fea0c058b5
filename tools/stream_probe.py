#!/usr/bin/env python
"""Robustness probe of the measured stream selection: time the ViT-S step after the process has created N unrelated HIP streams
(which shifts HIP's round-robin stream -> hardware-queue assignment).  usage: tools/stream_probe.py N [use_pool_stream]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pool = len(sys.argv) > 2 and sys.argv[2] == "1"
junk = [torch.cuda.Stream() for _ in range(n)]          # created BEFORE the library's streams
from bench import Args, make_batch
from fedcola_amd import _lib
from fedcola_amd.mome import create_model
a = Args(); a.precision = "bf16"
dev = torch.device("cuda"); torch.manual_seed(1)
model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
model.train()
B, seq = 64, a.seq_len
img, ids = make_batch(B, seq, a.vocab_size, 0, dev)
nn_ = model.flat.numel()
grads = torch.zeros(nn_, device=dev); m1 = torch.zeros(nn_, device=dev); m2 = torch.zeros(nn_, device=dev)
lossbuf = torch.zeros(2, device=dev)
model.prepare_weights(force=True); ws = model.workspace(B, seq)
L, P = _lib.lib(), _lib.ptr
cur = torch.cuda.Stream() if pool else torch.cuda.current_stream()
with torch.cuda.stream(cur):
    sp = _lib.stream_ptr()
    def step(k):
        _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), None,
                                    B, seq, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lossbuf), P(ws), ws.numel(), sp))
    for k in range(1, 6): step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(6, 36): step(k)
    torch.cuda.synchronize()
    print(f"junk streams {n} caller {'pool' if pool else 'default'}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step")
