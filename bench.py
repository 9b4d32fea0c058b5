#!/usr/bin/env python
"""bench.py -- img-txt pairs/sec of the FedCola client step (mome_small_patch16 = ViT-S image tower + same-width text
tower, B=64, 224x224 RGB, 32-token captions, AdamW) on N MI355X GPUs, one federated client per GPU.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run, one rank per GPU)

A "step" is one iteration of FedavgClient.update's batch loop (zero_grad, forward, contrastive loss, backward, AdamW) on a
synthetic Flickr30k-shaped batch already resident in HBM.  Clients are independent during local epochs (weak scaling);
the timed region ends with one FedAvg aggregation (RCCL all-reduce of the pre-weighted flat parameter buffer).
Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIR_GFLOP = 31.61           # algorithmic GEMM FLOPs per img-txt pair fwd+bwd, ViT-S, N_img=197, N_txt=32
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


class Args:
    vocab_size, seq_len, dropout = 7732, 32, 0.0
    shared_param, share_scope, colearn_param = "none", "dataset", "none"
    precision = "bf16"


def make_batch(B, seq, vocab, seed, device):
    g = torch.Generator().manual_seed(1000 + seed)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
    ids = torch.randint(1, vocab, (B, seq), generator=g)
    lens = torch.randint(8, seq + 1, (B,), generator=g)
    ids[torch.arange(seq)[None, :] >= lens[:, None]] = 0
    return img.to(device), ids.to(device)


def gemm_roofline(steps=50):
    """Dominant kernel: k_gemm_mfma (NT, fc1 shape M=12608 N=1536 K=384) timed with HIP events on its own stream."""
    from fedcola_amd import _lib
    L = _lib.lib()
    M, N, K = 64 * 197, 1536, 384
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    Cm = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    st = torch.cuda.Stream()
    sp = C.c_void_p(st.cuda_stream)
    P = _lib.ptr
    with torch.cuda.stream(st):
        for _ in range(5):
            _lib.check(L.fc_k_gemm(1, 0, 1, 1, P(A), P(W), P(Cm), M, N, K, P(bias), 0, sp))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps):
            _lib.check(L.fc_k_gemm(1, 0, 1, 1, P(A), P(W), P(Cm), M, N, K, P(bias), 0, sp))
        e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    flops = 2.0 * M * N * K
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01", "roofline_pmc.json")   # FETCH_SIZE (x2, gfx950) + WRITE_SIZE of this launch
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
    alg_bytes = 2.0 * (M * K + N * K + M * N)
    return dict(bound="mfma", kernel="k_gemm_mfma<NT,bias> fc1 shape 12608x1536x384 bf16 (dominant kernel family of the step)",
                achieved=round(achieved, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(achieved / PEAK_BF16_TFLOPS, 4),
                traffic=traffic, us_per_launch=round(ms * 1e3, 2), algorithmic_flops_per_launch=flops,
                algorithmic_bytes_per_launch=alg_bytes, hbm_frac=round(alg_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))


def cpu_baseline_child(B, seq, vocab, steps):
    """Runs in a CHILD process (no GPU touched): the oracle's explicit fp32 client step on the host cores."""
    from oracle import mome_oracle as O
    from fedcola_amd.mome import create_model
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    threads = max(1, min(cores, 64))          # torch CPU GEMMs stop scaling (and spin-wait badly) far below 256 threads
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    a = Args()
    a.precision = "fp32"
    m = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"])
    p = {k: v.clone() for k, v in m.state_dict().items()}
    cfg = O.OracleCfg(D=384, depth=12, heads=6, vocab=vocab, max_text_len=seq)
    img, ids = make_batch(B, seq, vocab, 0, "cpu")
    state = dict(step=0, m={}, v={})
    O.client_step(p, cfg, ("img+txt", img, ids), state, lr=1e-4)     # warm-up
    best = None
    for th in sorted({t for t in (16, 32, 64) if t <= max(16, threads)}):   # torch CPU GEMMs at these sizes do not scale to every core
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        O.client_step(p, cfg, ("img+txt", img, ids), state, lr=1e-4)
        d = time.perf_counter() - t0
        if best is None or d < best[0]:
            best = (d, th)
    threads = best[1]
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.client_step(p, cfg, ("img+txt", img, ids), state, lr=1e-4)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps(dict(value=round(B / dt, 2), unit="img-txt pairs/s", cores=threads, kind="port",
                          sample=f"{steps} timed + 1 warm-up fp32 steps of the same B={B} ViT-S workload by oracle/mome_oracle.py "
                                 f"(torch CPU ops, {threads} threads; host reports {os.cpu_count()} logical CPUs)")), flush=True)


def cpu_baseline(B, seq, vocab, steps=2, timeout=240):
    """Bounded CPU baseline in a child process started BEFORE this process touches the GPU."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--batch", str(B), "--steps", str(steps)],
                             capture_output=True, text=True, timeout=timeout, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if line else dict(value=None, unit="img-txt pairs/s", kind="port", sample="child failed: " + out.stderr[-300:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="img-txt pairs/s", kind="port", cores=None, sample=f"CPU baseline exceeded {timeout}s and was skipped")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="non-default: batches start in pinned host memory and reach the GPU through "
                    "fedcola_amd.loaders.DevicePrefetcher (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--fedprox-mu", type=float, default=0.0, help="non-default workload: FedproxClient step (proximal term, row N3)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_child:
        cpu_baseline_child(a.batch, Args.seq_len, Args.vocab_size, min(a.steps, 3))
        return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    cpu_base = None
    if world == 1 and not a.no_cpu_baseline:
        cpu_base = cpu_baseline(a.batch, Args.seq_len, Args.vocab_size)      # before any GPU call in this process
    if os.environ.get("FC_BENCH_ONE_DEVICE"):      # validation of the N>1 code path on a 1-GPU box: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("FC_BENCH_BACKEND", "nccl")               # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from fedcola_amd import _lib
    from fedcola_amd.mome import create_model
    args = Args()
    args.precision = a.precision
    torch.manual_seed(1 + rank)
    model = create_model("mome_small_patch16", False, args=args, num_classes=[None, None], modalities=["img", "txt"],
                         tasks=["rtv", "rtv"]).to(dev)
    model.train()
    B, seq = a.batch, args.seq_len
    img, ids = make_batch(B, seq, args.vocab_size, rank, dev)
    n = model.flat.numel()
    grads = torch.zeros(n, device=dev); m1 = torch.zeros(n, device=dev); m2 = torch.zeros(n, device=dev)
    lossbuf = torch.zeros(2, device=dev)
    model.prepare_weights(force=True)
    ws = model.workspace(B, seq)
    L, P = _lib.lib(), _lib.ptr
    sp = _lib.stream_ptr()
    step_no = [0]

    if a.fedprox_mu > 0:
        gflat = model.flat.detach().clone()
        pscr = torch.empty(L.fc_prox_scratch_bytes(model._handle.h), dtype=torch.uint8, device=dev)

    feed = None
    if a.h2d:
        from fedcola_amd.loaders import DevicePrefetcher
        himg, hids = img.cpu().pin_memory(), ids.cpu().pin_memory()      # what DataLoader(pin_memory=True) hands over

        def host_batches():
            while True:
                yield himg, hids
        feed = iter(DevicePrefetcher(host_batches(), dev, depth=2, stream=model.side_stream()))

    def step():
        nonlocal img, ids
        if feed is not None:
            img, ids = next(feed)
        step_no[0] += 1
        args = (model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), None,
                B, seq, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, step_no[0], P(lossbuf), P(ws), ws.numel(), sp)
        if a.fedprox_mu > 0:
            _lib.check(L.fc_client_step_prox(*args, P(gflat), a.fedprox_mu, P(pscr), pscr.numel()))
        else:
            _lib.check(L.fc_client_step(*args))

    # FedAvg aggregation of the `world` concurrent clients through the product path (fedcola_amd/aggregate.py): host-computed
    # coefficient table + closed-form weights -> one HIP blend kernel per rank -> one RCCL all-reduce over xGMI
    plan = None
    if world > 1:
        from fedcola_amd import aggregate as agg
        import copy
        cids = list(range(world))                                          # one client per rank
        keys = list(model.required_params().keys())
        coef = {k: {i: 1.0 / world for i in cids} for k in keys}           # equal client sizes, scope 'dataset'
        plan = agg.build_plan(model, cids, coef, {i: model.segments for i in cids})
        global_model = copy.deepcopy(model)

    def aggregate():
        if world > 1:
            agg.aggregate(global_model, plan, {rank: model.flat.data}, rank=rank, world=world)
            model.flat.data.copy_(global_model.flat.data)                  # next round's download(): device-to-device
            model._bump()
            model.prepare_weights(force=True)                              # ... and its bf16 compute weights

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    aggregate()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enq = time.perf_counter() - t0        # host time to enqueue the steps (launch-bound if close to dt)
    aggregate()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = float(lossbuf[1])
    if rank == 0:
        pairs = world * B * a.steps / dt
        out = dict(metric="img-txt pairs/sec per client round (ViT-S+BERT-mini)", value=round(pairs, 1), unit="img-txt pairs/s", n_gpus=world,
                   steps=a.steps, warmup=a.warmup, ms_per_step=round(dt / a.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype=a.precision, data="synthetic",
                   config=dict(workload="Flickr30k FedCola, 1 img-txt client per GPU, mome_small_patch16 (ViT-S + 12x384 text tower), "
                                        f"B={B}, 224x224 RGB, {seq}-token captions, vocab 7732, AdamW lr 1e-4, drop-path 0"
                                        + (f", FedProx mu={a.fedprox_mu}" if a.fedprox_mu > 0 else "")
                                        + (", batches from host memory through the device prefetcher" if a.h2d else ""),
                               global_batch=world * B, parallelism=f"{world} concurrent clients + RCCL FedAvg all-reduce"),
                   step_mfma_frac=round(pairs * PAIR_GFLOP / 1e3 / (world * PEAK_BF16_TFLOPS), 4), last_loss=round(loss, 4), enqueue_ms_per_step=round(t_enq / a.steps * 1e3, 3))
        if not a.no_roofline:
            out["roofline"] = gemm_roofline()
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
