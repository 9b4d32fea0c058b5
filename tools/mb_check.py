#!/usr/bin/env python
"""Micro-batch invariance check (development aid): one bf16 client step at an odd batch size with FC_MICROBATCH from the env;
prints the loss and a fingerprint of the gradient buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import product_util as PU
from synth import det_state_dict
from fedcola_amd.mome import ModalityAgnosticTransformer as M
kind = sys.argv[3] if len(sys.argv) > 3 else "img+txt"
if kind == "img+txt":
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=128, depth=3, num_heads=2, vocab_size=64, max_text_len=16)
else:       # image-only classifier with trained re-param linears: chains without a text tower, classification head before the fork
    mk = dict(modalities=["img", None], num_classes=[10, None], tasks=["cls", None], embed_dim=128, depth=3, num_heads=2, vocab_size=64, max_text_len=16,
              with_aux=True, aux_trained=True)
torch.manual_seed(0)
sd = det_state_dict({k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}, base_seed=5)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 17
g = torch.Generator().manual_seed(3)
img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
ids = torch.randint(1, 64, (B, 16), generator=g)
model = PU.build_product(mk, "bf16", sd); model.train()
y = (torch.arange(B) * 3 + 1) % 10
loss, grads, st = PU.product_step(model, kind, img, ids, y, 1e-4)
flat = torch.cat([v.reshape(-1).double() for v in grads.values()])
print("MB", os.environ.get("FC_MICROBATCH", "2"), "B", B, "loss %.7f" % loss, "gsum %.9e gl2 %.9e" % (float(flat.sum()), float(flat.norm())))
if len(sys.argv) > 2:
    torch.save({k: v.clone() for k, v in grads.items()}, sys.argv[2])
