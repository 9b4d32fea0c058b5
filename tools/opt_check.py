#!/usr/bin/env python
"""Fused-optimizer check (development aid + tests/test_gpu_model.py): one bf16 client step with FC_FUSED_OPT from the env; saves the
parameters, both moments, the bf16 compute weights and the last gradients.  usage: opt_check.py img+txt|img OUT.pt [width=128] [B]
width 128: every layer linear takes the 128x128-tile grouped kernel; width 384 (depth 2, B = 16: two micro-batch chains): the wide
128x384-tile kernels the ViT-S / ViT-B steps use (FC_DW_WIDE = 1 | 2 selects their two forms)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import product_util as PU
from synth import det_state_dict
from fedcola_amd.mome import ModalityAgnosticTransformer as M
kind = sys.argv[1]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 128
common = dict(embed_dim=128, depth=5, num_heads=2, vocab_size=64, max_text_len=16) if width == 128 else \
    dict(embed_dim=width, depth=2, num_heads=width // 64, vocab_size=64, max_text_len=16)
if kind == "img+txt":
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], **common)
else:
    mk = dict(modalities=["img", None], num_classes=[10, None], tasks=["cls", None], **common)
torch.manual_seed(0)
sd = det_state_dict({k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}, base_seed=5)
B = int(sys.argv[4]) if len(sys.argv) > 4 else (12 if width == 128 else 16)      # B >= 24 beside a text tower: three chains in both directions
g = torch.Generator().manual_seed(3)
img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
ids = torch.randint(1, 64, (B, 16), generator=g)
y = torch.arange(B) % 10
model = PU.build_product(mk, "bf16", sd); model.train()
# ONE step (number 3) from non-trivial moments: the step's inputs are identical in both runs, so every tensor the fused epilogue steps
# must come out bit-identical (a second step would see the atomics' run-to-run noise in the embedding / LayerNorm gradients)
n = model.flat.numel()
g2 = torch.Generator().manual_seed(11)
st = dict(grads=torch.zeros(n).cuda(), m=(torch.randn(n, generator=g2) * 1e-3).cuda(), v=(torch.rand(n, generator=g2) * 1e-5).cuda(), loss=torch.zeros(2).cuda())
loss, grads, st = PU.product_step(model, kind, img, ids, y, 1e-3, wd=0.01, step=3, state=st)
out = dict(p=model.flat.detach().cpu(), m=st["m"].cpu(), v=st["v"].cpu(), g=st["grads"].cpu(), wc=model._wc_or_flat().detach().cpu().view(torch.int16),
           loss=loss, segs={k: (int(v["offset"]), int(v["numel"])) for k, v in model.segments.items()})
print("FUSED", os.environ.get("FC_FUSED_OPT", "1"), kind, "loss %.7f" % loss, "psum %.9e" % float(out["p"].double().sum()))
torch.save(out, sys.argv[2])
