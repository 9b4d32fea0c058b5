#!/usr/bin/env python
"""Client-step time of the uni-modal clients of the mixed FedCola setting (SURVEY N3 / BASELINE config[4]) at the ViT-S width, bf16, B = 64:
an image classifier (CIFAR-100 shaped: 100 classes, 224x224) and a text classifier (AG_NEWS shaped: 4 classes, 40 tokens, vocab 30 522),
each with --aux --aux_trained re-param linears (the uni-modal clients carry them, fedavgclient.py:158-184).  A record, not the bench metric.
usage: tools/unimodal_step.py [steps]"""
import json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from fedcola_amd import _lib
from fedcola_amd.mome import ModalityAgnosticTransformer as M
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 64
L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr()
common = dict(embed_dim=384, depth=12, num_heads=6, vocab_size=30522, max_text_len=40, with_aux=True, aux_trained=True, precision="bf16")
out = []
for kind in ("img", "txt"):
    if kind == "img":
        mk = dict(modalities=["img", None], num_classes=[100, None], tasks=["cls", None], **common)
    else:
        mk = dict(modalities=[None, "txt"], num_classes=[None, 4], tasks=[None, "cls"], **common)
    torch.manual_seed(0)
    model = M(**mk).cuda(); model.train()
    g = torch.Generator().manual_seed(1)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1).cuda() if kind == "img" else None
    ids = torch.randint(1, 30522, (B, 40), generator=g).cuda() if kind == "txt" else None
    y = (torch.arange(B) % (100 if kind == "img" else 4)).cuda()
    n = model.flat.numel()
    grads, m1, m2 = (torch.zeros(n, device="cuda") for _ in range(3)); loss = torch.zeros(2, device="cuda")
    model.prepare_weights(force=True)
    n_txt = 40 if kind == "txt" else 0
    ws = model.workspace(B, n_txt)
    def step(i):
        _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), P(y), B, n_txt, None,
                                    1e-4, 0.9, 0.999, 1e-8, 0.0, i, P(loss), P(ws), ws.numel(), sp))
    for i in range(1, 6): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6, 6 + steps): step(i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    out.append(dict(client=kind + "-cls, ViT-S width, --aux --aux_trained, bf16", B=B, ms_per_step=round(ms, 3), samples_per_s=round(B / ms * 1e3, 1), params=int(n)))
    del model, ws, grads, m1, m2
print(json.dumps(out))
