#!/usr/bin/env python
"""Host batch assembly: torch's single-process DataLoader (the reference's loader) vs fedcola_amd.loaders.PinnedBatchLoader on a
Flickr30k-shaped tensor dataset (B = 64, 3x224x224 fp32 + 32 tokens).  usage: tools/loader_bench.py [n_samples=1024]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd.loaders import PinnedBatchLoader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


class DS(torch.utils.data.Dataset):
    def __init__(self):
        self.img = torch.randn(n, 3, 224, 224)
        self.ids = torch.randint(0, 7732, (n, 32))

    def __len__(self):
        return n

    def __getitem__(self, i):
        return self.img[i], self.ids[i], i // 5, i, i


class DSB(DS):
    def get_batch(self, idxs):
        i = torch.as_tensor(idxs)
        return self.img[i], self.ids[i], i // 5, i, i


ds = DS()
dsb = DSB()
out = {}
for name, mk in (("DataLoader", lambda: torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True)),
                 ("PinnedBatchLoader", lambda: PinnedBatchLoader(ds, 64, shuffle=True, workers=4)),
                 ("PinnedBatchLoader+get_batch", lambda: PinnedBatchLoader(dsb, 64, shuffle=True, workers=4))):
    list(zip(range(2), mk()))
    t0 = time.perf_counter()
    nb = sum(1 for _ in mk())
    out[name] = round((time.perf_counter() - t0) / nb * 1e3, 2)
print(json.dumps(dict(ms_per_batch=out, batch="64 x (3x224x224 fp32 + 32 int64 tokens)", host_threads=torch.get_num_threads())))
# end to end: the ViT-S client step fed by each loader through DevicePrefetcher (host batches -> H2D on the library's side stream)
if torch.cuda.is_available():
    from bench import Args
    from fedcola_amd import _lib
    from fedcola_amd.loaders import DevicePrefetcher
    from fedcola_amd.mome import create_model
    a = Args(); a.precision = "bf16"
    dev = torch.device("cuda")
    model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
    model.train()
    nn_ = model.flat.numel()
    grads = torch.zeros(nn_, device=dev); m1 = torch.zeros(nn_, device=dev); m2 = torch.zeros(nn_, device=dev); lossbuf = torch.zeros(2, device=dev)
    model.prepare_weights(force=True); ws = model.workspace(64, 32)
    L, P = _lib.lib(), _lib.ptr
    e2e = {}
    for name, mk in (("DataLoader", lambda: torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True, drop_last=True)),
                     ("PinnedBatchLoader", lambda: PinnedBatchLoader(ds, 64, shuffle=True, drop_last=True, workers=4, ahead=2)),
                     ("PinnedBatchLoader+get_batch", lambda: PinnedBatchLoader(dsb, 64, shuffle=True, drop_last=True, workers=4, ahead=2))):
        k = 0
        t0 = None
        for ep in range(2):
            for img, ids, *_ in DevicePrefetcher(mk(), dev, depth=2, stream=model.side_stream()):
                k += 1
                _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img.float()), P(ids), None, 64, 32,
                                            None, 1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lossbuf), P(ws), ws.numel(), _lib.stream_ptr()))
            if ep == 0:
                torch.cuda.synchronize(); t0 = time.perf_counter(); k0 = k
        torch.cuda.synchronize()
        e2e[name] = round((time.perf_counter() - t0) / (k - k0) * 1e3, 2)
    print(json.dumps(dict(client_loop_ms_per_step=e2e, note="ViT-S bf16 fused step fed from host memory; the device step alone is bench.py's ms_per_step (4.8 ms in round 3)")))
