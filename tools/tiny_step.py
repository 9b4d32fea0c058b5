#!/usr/bin/env python
"""Client-step time of BASELINE.json config[0]'s model on the device: mome_tiny_patch16 (ViT-Tiny: 192 wide, 12 layers, 3 heads), img-only
classification client (100 classes, CE loss), bf16, B = 64.  A record beside the parity test (tests/test_gpu_bf16_parity.py), not the bench
metric.   usage: tools/tiny_step.py [steps]"""
import json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from fedcola_amd import _lib
from fedcola_amd.mome import ModalityAgnosticTransformer as M
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr()
out = []
for B in (64, 16):
    mk = dict(modalities=["img", None], num_classes=[100, None], tasks=["cls", None], embed_dim=192, depth=12, num_heads=3, vocab_size=7732,
              max_text_len=32, precision="bf16")
    torch.manual_seed(0)
    model = M(**mk).cuda(); model.train()
    g = torch.Generator().manual_seed(1)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1).cuda()
    y = (torch.arange(B) % 100).cuda()
    n = model.flat.numel()
    grads, m1, m2 = (torch.zeros(n, device="cuda") for _ in range(3)); loss = torch.zeros(2, device="cuda")
    model.prepare_weights(force=True)
    ws = model.workspace(B, 0)

    def step(i):
        _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), None, P(y), B, 0, None,
                                    1e-4, 0.9, 0.999, 1e-8, 0.0, i, P(loss), P(ws), ws.numel(), sp))
    for i in range(1, 6): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6, 6 + steps): step(i)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
    gflop = 3 * 2.856 - 0.058          # fwd+bwd GEMM GFLOP per image, ViT-Tiny (pair 8.51 incl. the text tower; image tower only here)
    out.append(dict(client="img-cls, mome_tiny_patch16 (ViT-Tiny 192 x 12 x 3 heads), 100 classes, bf16", B=B, ms_per_step=round(ms, 3),
                    samples_per_s=round(B / ms * 1e3, 1), params=int(n)))
    del model, ws, grads, m1, m2
print(json.dumps(out))
