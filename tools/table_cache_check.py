#!/usr/bin/env python
"""Device-table cache check (tests/test_gpu_model.py): client steps whose batch size keeps changing, so that new weight-gradient /
LayerNorm-reduction / optimizer tables keep being cached; with FC_TABLE_CACHE_MAX=3 (tools build) the cache starts over several times
per run -- also in the middle of a step -- and every handle-side use of a table must survive that.  Saves the final parameters.
usage: table_cache_check.py OUT.pt"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import product_util as PU
from synth import det_state_dict
from fedcola_amd.mome import ModalityAgnosticTransformer as M
mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=384, depth=2, num_heads=6, vocab_size=64, max_text_len=16)
torch.manual_seed(0)
sd = det_state_dict({k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}, base_seed=5)
model = PU.build_product(mk, "bf16", sd); model.train()
st = None
losses = []
for step, B in enumerate([16, 12, 16, 9, 12, 16, 9, 16], start=1):
    g = torch.Generator().manual_seed(100 + step)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
    ids = torch.randint(1, 64, (B, 16), generator=g)
    loss, grads, st = PU.product_step(model, "img+txt", img, ids, None, 1e-3, wd=0.01, step=step, state=st)
    losses.append(loss)
torch.cuda.synchronize()
print("LIMIT", os.environ.get("FC_TABLE_CACHE_MAX", "default"), " ".join("%.6f" % l for l in losses))
torch.save(dict(p=model.flat.detach().cpu(), losses=losses), sys.argv[1])
