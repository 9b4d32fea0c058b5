#!/usr/bin/env python
"""fp32 mode, 768-wide 2-layer test model: the distance of the FORWARD features from the exact (fp64) ones, library vs the fp32 oracle,
and of the contrastive loss's feature gradients.  (Tools only: which side of the step carries the parity test's residual.)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import product_util as PU
from oracle import mome_oracle as O
from synth import det_state_dict
from fedcola_amd.mome import ModalityAgnosticTransformer as M
import test_gpu_fullsize as T
mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], **T.MKB)
cfg = O.OracleCfg(D=768, depth=2, heads=12, vocab=30522, max_text_len=40)
torch.manual_seed(2)
shapes = {k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}
sd = det_state_dict(shapes, base_seed=41)
img, ids = T._batch(8, 40, 30522)
p32 = {k: v.clone() for k, v in sd.items()}
p64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
o32, _ = O.forward(p32, cfg, [img, ids], feat_out=True)
o64, _ = O.forward(p64, cfg, [img.double(), ids], feat_out=True)
model = PU.build_product(mk, "fp32", sd); model.train()
with torch.no_grad():
    ol = model([img.cuda(), ids.cuda()], feat_out=True)
for i, nm in enumerate(("image features", "text features")):
    el = (ol[i].double().cpu() - o64[i]).abs(); eo = (o32[i].double() - o64[i]).abs()
    print(f"{nm}: |f| max {float(o64[i].abs().max()):.3f}; library max err {float(el.max()):.2e} rms {float(el.pow(2).mean().sqrt()):.2e} | fp32 oracle max {float(eo.max()):.2e} rms {float(eo.pow(2).mean().sqrt()):.2e}")
l64, da64, db64 = O.contrastive_loss(o64[0], o64[1])
for nm, (a, b) in (("library", (ol[0].cpu(), ol[1].cpu())), ("fp32 oracle", (o32[0], o32[1]))):
    l, da, db = O.contrastive_loss(a.double(), b.double())      # the exact loss gradient AT the inexact features
    print(f"{nm}: loss {float(l):.8f} (exact {float(l64):.8f}); d loss / d features at these features vs at the exact ones: rel max {float((da - da64).abs().max() / da64.abs().max()):.2e}"
          f" {float((db - db64).abs().max() / db64.abs().max()):.2e}")
