import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from fedcola_amd.mome import ModalityAgnosticTransformer as M
from synth import det_ids, det_tensor
torch.manual_seed(3)
m = M(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], img_size=32, patch_size=16, embed_dim=64, depth=2,
      num_heads=2, mlp_ratio=2, vocab_size=97, max_text_len=8, precision="fp32").cuda()
m.eval()
n_img, caps = 12, 5
images = det_tensor([n_img, 3, 32, 32], 11, 1.0)
tokens = det_ids([n_img * caps, 8], 5, 97)
perm = (torch.arange(n_img * caps) * 7 + 3) % (n_img * caps)
fi = torch.zeros(60, 64); ft = torch.zeros(60, 64)
for s in range(0, 60, 16):
    j = perm[s:s + 16]
    with torch.no_grad():
        o = m([images[j // caps].cuda(), tokens[j].cuda()], feat_out=True)
    fi[j] = o[0].float().cpu().reshape(len(j), -1); ft[j] = o[1].float().cpu().reshape(len(j), -1)
# duplicates: captions with equal tokens, images with equal index
worst_t = 0.0; worst_i = 0.0
for a in range(60):
    for b in range(a + 1, 60):
        if torch.equal(tokens[a], tokens[b]): worst_t = max(worst_t, float((ft[a] - ft[b]).abs().max()))
        if a // caps == b // caps: worst_i = max(worst_i, float((fi[a] - fi[b]).abs().max()))
print("duplicate captions: max feature difference", worst_t, " same image in different batches:", worst_i)
