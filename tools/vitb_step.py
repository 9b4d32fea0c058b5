#!/usr/bin/env python
"""Client-step time of BASELINE.json config[3]'s model (ViT-B/16 + BERT-base width: 768 wide, 12 layers, 12 heads, vocab 30 522,
40-token captions), img+txt client, bf16 -- a record next to the headline ViT-S line, not the bench metric.
usage: tools/vitb_step.py [B] [steps]"""
import json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import product_util as PU
from fedcola_amd import _lib
from fedcola_amd.mome import ModalityAgnosticTransformer as M
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=768, depth=12, num_heads=12,
          vocab_size=30522, max_text_len=40)
torch.manual_seed(0)
model = M(precision="bf16", **mk).cuda(); model.train()
g = torch.Generator().manual_seed(1)
img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1).cuda()
ids = torch.randint(1, 30522, (B, 40), generator=g).cuda()
n = model.flat.numel()
grads, m1, m2 = (torch.zeros(n, device="cuda") for _ in range(3))
loss = torch.zeros(2, device="cuda")
model.prepare_weights(force=True)
ws = model.workspace(B, 40)
L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr()
def step(i):
    _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), None, B, 40, None,
                                1e-4, 0.9, 0.999, 1e-8, 0.0, i, P(loss), P(ws), ws.numel(), sp))
for i in range(1, 6): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(6, 6 + steps): step(i)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
D, Hd, N, Nt, depth = 768, 3072, 197, 40, 12
flops_pair = 3 * 2 * depth * ((N + Nt) * (4 * D * D + 2 * D * Hd) + 2 * (N * N + Nt * Nt) * D) + 3 * 2 * (N - 1) * 768 * D
print(json.dumps(dict(model="ViT-B/16 + 12x768 text tower (vocab 30522, 40 tokens), img+txt, bf16", B=B, ms_per_step=round(ms, 3),
                      pairs_per_s=round(B / ms * 1e3, 1), params=int(n), step_mfma_frac=round(flops_pair * B / (ms * 1e-3) / 2.5e15, 4))))
