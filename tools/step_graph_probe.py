#!/usr/bin/env python
"""Host time of fc_client_step per call, step by step (whole-step graph: eager, eager, capture, replays): tools build,
FC_STEP_GRAPH=0|1 python tools/step_graph_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FC_PROBES_LIB", "1")
import torch
import bench
from fedcola_amd import _lib
from fedcola_amd.mome import create_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
args = bench.Args(); args.precision = "bf16"
dev = torch.device("cuda", 0)
model = create_model("mome_small_patch16", False, args=args, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev); model.train()
B, seq = 64, args.seq_len
img, ids = bench.make_batch(B, seq, args.vocab_size, 0, dev)
n = model.flat.numel()
g, m1, m2, lb = (torch.zeros(n, device=dev) for _ in range(3)) .__iter__().__next__(), None, None, None
g = torch.zeros(n, device=dev); m1 = torch.zeros(n, device=dev); m2 = torch.zeros(n, device=dev); lb = torch.zeros(2, device=dev)
model.prepare_weights(force=True); ws = model.workspace(B, seq)
L, P, sp = _lib.lib(), _lib.ptr, _lib.stream_ptr()
host, gpu = [], []
for k in range(1, steps + 1):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(g), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), None, B, seq, None,
                                1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lb), P(ws), ws.numel(), sp))
    host.append((time.perf_counter() - t0) * 1e3)
    e1.record(); e1.synchronize()
    gpu.append(e0.elapsed_time(e1))
print("FC_STEP_GRAPH =", os.environ.get("FC_STEP_GRAPH", "(default)"))
print("host ms per call:", " ".join(f"{x:.2f}" for x in host))
print("gpu  ms per call:", " ".join(f"{x:.2f}" for x in gpu))
