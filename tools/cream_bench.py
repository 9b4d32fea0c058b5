#!/usr/bin/env python
"""Time CreamFL's public-set distillation step on the ViT-S img+txt model next to the plain fused client step.
    python tools/cream_bench.py [pub_samples=512] [pub_batch=64]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import Args, make_batch
from fedcola_amd.client.creamflclient import CreamflClient
from fedcola_amd.mome import create_model

P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
a = Args(); a.precision = "bf16"
for k, v in dict(E=1, B=64, lr=1e-4, optimizer="AdamW", no_shuffle=True, interintra_weight=0.5, pub_batch_size=PB, no_mm_contrastive=False,
                 max_grad_norm=0.0, debug=False, distributed=False, mm_distributed=False, weight_decay=0.0, prefetch=False).items():
    setattr(a, k, v)
torch.manual_seed(0)
dev = torch.device("cuda")
model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
img, ids = make_batch(PB, a.seq_len, a.vocab_size, 0, dev)
himg, hids = img.cpu(), ids.cpu()


class Pub(torch.utils.data.Dataset):
    def __len__(self): return P
    def __getitem__(self, i): return himg[i % PB], hids[i % PB], i, i, i


class Train(torch.utils.data.Dataset):
    def __len__(self): return 64 * 8
    def __getitem__(self, i): return himg[i % PB], hids[i % PB], i // 5, i, i


cl = CreamflClient(args=a, training_set=Train(), test_set=Train(), task="rtv", modality="img+txt", eval_metrics=[], criterion="ContrastiveLoss")
cl.id, cl.dataset, cl.device, cl.pub_dataset = 0, "Flickr30k", "cuda", Pub()
g = torch.nn.functional.normalize(torch.randn(P, 384, device=dev), dim=-1)
cl.global_img_feature, cl.global_txt_feature, cl.distill_index = g, g.flip(0).contiguous(), list(range(P))
cl.model = model
# pre-collated batches: the single-process DataLoader (38.5 MB collate per batch, ~100 ms) is the host side's business (row N4)
pub_batches = [(himg, hids, torch.arange(PB), torch.arange(PB), torch.arange(b * PB, (b + 1) * PB)) for b in range(P // PB)]
cl.get_pub_loader = lambda dataset, batch_size=PB: pub_batches
cl.train_loader = [(himg.pin_memory(), hids.pin_memory(), torch.arange(PB), torch.arange(PB), torch.arange(PB)) for _ in range(8)]
times = {}
orig = cl._after_epoch


def timed(e, st, step):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(e, st, step)
    torch.cuda.synchronize(); times["distill"] = (time.perf_counter() - t0) / (P // PB)
    return r
cl._after_epoch = timed
torch.cuda.synchronize(); t0 = time.perf_counter()
cl.update()
torch.cuda.synchronize(); tot = time.perf_counter() - t0
local = (tot - times["distill"] * (P // PB)) / 8
print(json.dumps(dict(model="mome_small_patch16 img+txt bf16", pub_samples=P, pub_batch=PB, distill_ms_per_step=round(times["distill"] * 1e3, 2),
                      local_ms_per_step=round(local * 1e3, 2),
                      note="distill step = old-model forward + forward + moon/inter losses + backward + clip_grad_norm + per-segment AdamW, host-synchronised per batch")))
