import sys, os, ctypes as C, math
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_gpu_bf16_parity as T
import product_util as PU
from oracle import mome_oracle as O
from fedcola_amd.mome import ModalityAgnosticTransformer as M
depth, B, D, H, Hd, seq = 1, 16, 384, 6, 1536, 32
mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=D, depth=depth, num_heads=H, vocab_size=7732, max_text_len=seq)
torch.manual_seed(5)
sd = {k: v.clone() for k, v in M(**mk).state_dict().items()}
g = torch.Generator().manual_seed(9)
for k in sd:
    if "pos_embed" in k or "cls_token" in k: sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
img, ids = T._batch(B, seq, 7732)
cfg = O.OracleCfg(D=D, depth=depth, heads=H, vocab=7732, max_text_len=seq)
p = {k: v.clone() for k, v in sd.items()}
with O.emulate_bf16():
    outs_o, cache = O.forward(p, cfg, [img, ids], feat_out=True)
model = PU.build_product(mk, "bf16", sd); model.train()
with torch.no_grad():
    outs = model([img.cuda(), ids.cuda()], feat_out=True)
torch.cuda.synchronize()
ws = model._ws
off = [0]
def take(nbytes):
    o = off[0]; off[0] += (nbytes + 255) // 256 * 256; return o
def bf(o, shape):
    n = 1
    for s_ in shape: n *= s_
    return ws[o:o + 2 * n].view(torch.bfloat16).float().cpu().reshape(shape)
take(32)
N = 197; Mr = B * N
o_patches = take(B * 196 * 768 * 2); take(B * 196 * D * 2)
ox = [take(Mr * D * 2) for _ in range(depth + 1)]
[take(Mr * D * 2) for _ in range(depth + 1)]
for _ in range(4): take(4 * Mr)
take(4 * B * H * N)
o_h1 = take(Mr * D * 2); o_qkv = take(Mr * 3 * D * 2); o_o = take(Mr * D * 2); o_xmid = take(Mr * D * 2); o_h2 = take(Mr * D * 2)
o_u = take(Mr * Hd * 2); o_gact = take(Mr * Hd * 2)
tc = cache["towers"][0]; bc = tc["blocks"][0]
def cmp(name, got, ref):
    d = got - ref
    print(f"{name:8s} rel L2 {float(d.norm()/ref.norm()):.2e}  max abs {float(d.abs().max()):.3e}  mismatching {int((d != 0).sum())}/{d.numel()}")
pt = bf(o_patches, (B, 196, 768)); cmp("patches", pt, tc["patches"])
x0 = bf(ox[0], (B, N, D))
# oracle x0: recompute
Wp = O.R if False else None
with O.emulate_bf16():
    ptr_ = O.R(O.patchify(img, 16)); Wp = O.R(p["embeddings.0.embed.proj.weight"].reshape(D, -1))
    tok = ptr_ @ Wp.t() + p["embeddings.0.embed.proj.bias"]
    h0 = O.R(torch.cat([p["embeddings.0.cls_token"].expand(B, -1, -1), tok], 1) + p["embeddings.0.pos_embed"])
cmp("x0", x0, h0)
cmp("h1", bf(o_h1, (B, N, D)), bc["h1"])
qkv = bf(o_qkv, (B, N, 3, H, 64)).permute(2, 0, 3, 1, 4)
cmp("q", qkv[0] * 0.125, bc["q"]); cmp("k", qkv[1], bc["k"]); cmp("v", qkv[2], bc["v"])
cmp("o", bf(o_o, (B, N, D)), bc["O"])
cmp("h2", bf(o_h2, (B, N, D)), bc["h2"])
cmp("gp", bf(o_u, (B, N, Hd)), bc["gp"])
cmp("gact", bf(o_gact, (B, N, Hd)), bc["gact"])
cmp("x1", bf(ox[1], (B, N, D)), None if False else (lambda: None) and bc["h2"] * 0 + bf(ox[1], (B, N, D)))  # placeholder
print("outs err", [float((o.cpu() - oo).abs().max()) for o, oo in zip(outs, outs_o)])
