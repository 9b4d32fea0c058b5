#!/usr/bin/env python
"""Where a FedavgClient round spends its time beside the device steps (development aid): wall-clock stamps inside update() through a
wrapped loader / fc_client_step, for bench.py's client_round workload.  usage: tools/client_round_trace.py"""
import os, sys, time, json, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fedcola_amd import _lib, aggregate as agg
from fedcola_amd.mome import create_model
from fedcola_amd.client.fedavgclient import FedavgClient
from fedcola_amd.loaders import prefetch as PF
a = bench.Args(); a.precision = "bf16"
dev = torch.device("cuda")
model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
B, seq, nsteps = 64, 32, 20


class CArgs: pass
ca = CArgs()
ca.__dict__.update(dict(vocab_size=a.vocab_size, seq_len=seq, dropout=0.0, optimizer="AdamW", lr=1e-4, weight_decay=0.0, E=1, B=B, no_shuffle=False, debug=False,
                        with_aux=False, aux_attn_only=False, aux_mlp_only=False, max_grad_norm=0.0, distributed=False, mm_distributed=False, train_only=True))
ds = bench.InMemoryPairs(nsteps * B, seq, a.vocab_size)
client = FedavgClient(ca, ds, ds, task="rtv", eval_metrics=[], modality="img+txt", criterion="ContrastiveLoss")
client._BaseClient__identifier = 0
client.dataset = "Flickr30k"
gmodel = copy.deepcopy(model)
stamps = []
evs = []
L = _lib.lib()
orig_step = L.fc_client_step


class Wrap:      # records when each step is handed to the library and when the call returns
    def __call__(self, *args):
        t0 = time.perf_counter(); r = orig_step(*args); ev = torch.cuda.Event(enable_timing=True); ev.record(); evs.append(ev)
        stamps.append(("step", t0, time.perf_counter())); return r
L.fc_client_step = Wrap()
orig_iter = PF.DevicePrefetcher.__iter__


def traced_iter(self):
    for b in orig_iter(self):
        stamps.append(("batch", time.perf_counter(), 0)); yield b
PF.DevicePrefetcher.__iter__ = traced_iter
for r in range(4):
    del stamps[:]; del evs[:]
    torch.cuda.synchronize(); ev0 = torch.cuda.Event(enable_timing=True); ev0.record(); t0 = time.perf_counter()
    client.download({"Flickr30k": gmodel}); t1 = time.perf_counter()
    res = client.update(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    steps = [s for s in stamps if s[0] == "step"]; batches = [s for s in stamps if s[0] == "batch"]
    if r >= 2:
        print(json.dumps(dict(round=r, download_ms=round((t1 - t0) * 1e3, 2), update_ms=round((t2 - t1) * 1e3, 2),
                              first_batch_at_ms=round((batches[0][1] - t1) * 1e3, 2), first_step_call_ms=round((steps[0][2] - steps[0][1]) * 1e3, 2),
                              second_step_call_ms=round((steps[1][2] - steps[1][1]) * 1e3, 2), median_step_call_ms=round(sorted(s[2] - s[1] for s in steps)[len(steps) // 2] * 1e3, 2),
                              last_step_returned_at_ms=round((steps[-1][2] - t1) * 1e3, 2), device_step_end_ms_after_round_start=[round(ev0.elapsed_time(e), 1) for e in evs],
                              device_step_ms=[round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(len(evs) - 1)], batch_gaps_ms=[round((batches[i + 1][1] - batches[i][1]) * 1e3, 1) for i in range(len(batches) - 1)])))
