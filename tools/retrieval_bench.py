#!/usr/bin/env python
"""Time the retrieval-evaluation kernels at Flickr30k-test and COCO-5k sizes against the CPU oracle.

    python tools/retrieval_bench.py [reps]

Prints one JSON line per size: t2i + i2t ranking time (HIP events), the fp64-GEMM rate, the CPU oracle's time on a bounded
query sample scaled to the full query count."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from fedcola_amd import _lib
from oracle import retrieval_oracle as ro

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = _lib.lib(); P = _lib.ptr
for name, n_img, caps, d in (("flickr30k_test_1k", 1000, 5, 384), ("coco_5k", 5000, 5, 384)):
    gen = torch.Generator().manual_seed(1)
    img = torch.nn.functional.normalize(torch.randn(n_img, d, generator=gen), dim=-1)
    cap = torch.nn.functional.normalize(img.repeat_interleave(caps, 0) + 0.35 * torch.randn(n_img * caps, d, generator=gen), dim=-1)
    li = torch.arange(n_img, dtype=torch.int64); lc = li.repeat_interleave(caps)
    qi, qc = img.double().cuda(), cap.double().cuda()
    lid, lcd = li.cuda(), lc.cuda()
    qb = 1024
    scratch = torch.empty(L.fc_retrieval_scratch_bytes(qb, max(n_img, n_img * caps)), dtype=torch.uint8, device="cuda")
    o1 = torch.empty(n_img * caps, dtype=torch.int64, device="cuda"); o2 = torch.empty(n_img, dtype=torch.int64, device="cuda")
    sp = _lib.stream_ptr()

    def run():
        _lib.check(L.fc_retrieval_best_ranks(P(qc), P(qi), P(lcd), P(lid), n_img * caps, n_img, d, P(scratch), scratch.numel(), P(o1), sp))
        _lib.check(L.fc_retrieval_best_ranks(P(qi), P(qc), P(lid), P(lcd), n_img, n_img * caps, d, P(scratch), scratch.numel(), P(o2), sp))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2 * 2.0 * n_img * n_img * caps * d
    sbytes = 2 * 3 * 8.0 * n_img * n_img * caps            # similarity matrix written once, read twice, both directions
    # CPU oracle on a bounded sample of queries (all gallery items), scaled
    ns = min(64, n_img)
    t0 = time.perf_counter()
    r_c = ro.best_ranks(cap[:ns * caps].double().numpy(), img.double().numpy(), lc[:ns * caps].numpy().astype(float), li.numpy().astype(float))
    r_i = ro.best_ranks(img[:ns].double().numpy(), cap.double().numpy(), li[:ns].numpy().astype(float), lc.numpy().astype(float))
    cpu_s = (time.perf_counter() - t0) * n_img / ns
    assert np.array_equal(o1[:ns * caps].cpu().numpy(), r_c.astype(np.int64)) and np.array_equal(o2[:ns].cpu().numpy(), r_i.astype(np.int64))
    print(json.dumps(dict(case=name, queries=n_img * (caps + 1), gallery=[n_img, n_img * caps], d=d, hip_ms=round(ms, 3),
                          fp64_gemm_tflops=round(flops / ms / 1e9, 2), sims_GBps=round(sbytes / ms / 1e6, 1),
                          cpu_oracle_s_scaled=round(cpu_s, 2), cpu_sample_queries=ns * (caps + 1), cpu_threads=torch.get_num_threads(),
                          speedup=round(cpu_s * 1e3 / ms, 1))))
