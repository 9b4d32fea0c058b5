#!/usr/bin/env python
"""fp32 mode, 768-wide 2-layer test model, one img+txt step: every backward intermediate the workspace keeps (loss gradient, the block's dY
tensors, the residual-stream gradient per layer) against the EXACT (fp64) oracle, beside the fp32 oracle's own distance.  The first tensor
where the library's figure exceeds the fp32 oracle's names the kernel that carries the parity test's residual.  (Tools only.)"""
import ctypes as C, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import product_util as PU
from oracle import mome_oracle as O
from synth import det_state_dict
from fedcola_amd import _lib
from fedcola_amd.mome import ModalityAgnosticTransformer as M
import test_gpu_fullsize as T
mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], **T.MKB)
cfg = O.OracleCfg(D=768, depth=2, heads=12, vocab=30522, max_text_len=40)
torch.manual_seed(2)
shapes = {k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}
sd = det_state_dict(shapes, base_seed=41)
B, seq, D = 8, 40, 768
img, ids = T._batch(B, seq, 30522)
blk = O.block_bwd


def run(p, im):
    taps = {}

    def rec(p_, pre, dx2, *a, **kw):
        t = {}
        out = blk(p_, pre, dx2, *a, tap=t, **kw)
        t["gx_out"] = dx2; t["gx_in"] = out
        taps[pre] = t
        return out
    O.block_bwd = rec
    try:
        outs, cache = O.forward(p, cfg, [im, ids], feat_out=True)
        loss, da, db = O.contrastive_loss(outs[0], outs[1])
        O.backward(p, cfg, cache, [da, db])
    finally:
        O.block_bwd = blk
    taps["dout"] = [da, db]
    return taps
t32 = run({k: v.clone() for k, v in sd.items()}, img)
t64 = run({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}, img.double())
model = PU.build_product(mk, "fp32", sd); model.train()
PU.product_step(model, "img+txt", img, ids, None, 1e-4)


def ws(tower, layer, name, shape):
    off, nb = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib().fc_workspace_tensor(model._handle.h, B, seq, tower, layer, name.encode(), C.byref(off), C.byref(nb)))
    return model._ws[off.value: off.value + nb.value].view(torch.float32).cpu().reshape(shape)


def fig(name, lib, o32, o64):
    mx = float(o64.abs().max())
    el, eo = (lib.double() - o64).abs(), (o32.double() - o64).abs()
    print(f"{name:34s} max |exact| {mx:.2e} | library max {float(el.max()) / mx:.2e} rms {float(el.pow(2).mean().sqrt()) / mx:.2e} | fp32 oracle max {float(eo.max()) / mx:.2e} rms {float(eo.pow(2).mean().sqrt()) / mx:.2e}"
          f" | ratio max {float(el.max() / eo.max()):.1f} rms {float(el.pow(2).mean().sqrt() / eo.pow(2).mean().sqrt()):.1f}")


for tower, N in ((0, 197), (1, seq)):
    print(f"# tower {tower} ({'image' if tower == 0 else 'text'}, {B} x {N} rows)")
    fig("d loss / d features", ws(tower, 0, "dout", (B, D)), t32["dout"][tower], t64["dout"][tower])
    for l in (1, 0):
        pre = f"blockses.{tower}.{l}"
        a, b = t32[pre], t64[pre]
        fig(f"layer {l} gx_out (incoming)", ws(tower, l + 1, "gx", (B, N, D)), a["gx_out"], b["gx_out"])
        fig(f"layer {l} du  (fc2 dX x gelu')", ws(tower, l, "gdu", (B, N, 4 * D)), a["du"], b["du"])
        fig(f"layer {l} dx1 (+ LN2 bwd)", ws(tower, l, "gxmid", (B, N, D)), a["dx1"], b["dx1"])
        fig(f"layer {l} dqkv (attention bwd)", ws(tower, l, "gdqkv", (B, N, 3 * D)), a["dqkv"], b["dqkv"])
        fig(f"layer {l} gx_in (+ LN1 bwd)", ws(tower, l, "gx", (B, N, D)), a["gx_in"], b["gx_in"])

# ---- teacher-forced: the stage dqkv -> (qkv dX GEMM) -> (LayerNorm-1 backward) -> + dx1, on the LIBRARY's own inputs, in fp64 and in torch fp32
print("# teacher-forced stage gx_in = dx1 + LN1_bwd(dqkv . Wqkv) on the library's own dqkv / dx1 / x / mean / rstd: the error the stage itself adds")
for tower, N in ((0, 197), (1, seq)):
    for l in (1, 0):
        pre = f"blockses.{tower}.{l}"
        dqkv = ws(tower, l, "gdqkv", (B * N, 3 * D)); dx1 = ws(tower, l, "gxmid", (B * N, D)); x = ws(tower, l, "x", (B * N, D))
        mean = ws(tower, l, "mean1", (B * N, 1)); rstd = ws(tower, l, "rstd1", (B * N, 1))
        W = sd[pre + ".attn.qkv.weight"]; gam = sd[pre + ".norm1.weight"]

        def stage(dt):
            dh = dqkv.to(dt) @ W.to(dt)
            xh = (x.to(dt) - mean.to(dt)) * rstd.to(dt)
            gdy = dh * gam.to(dt)
            dxn = rstd.to(dt) * (gdy - gdy.mean(1, keepdim=True) - xh * (gdy * xh).mean(1, keepdim=True))
            return dh, dx1.to(dt) + dxn
        dh64, g64 = stage(torch.float64); dh32, g32 = stage(torch.float32)
        got = ws(tower, l, "gx", (B * N, D))
        mx = float(g64.abs().max())
        el, eo = (got.double() - g64).abs(), (g32.double() - g64).abs()
        print(f"tower {tower} layer {l}: max |gx_in| {mx:.2e}, max |dh1| {float(dh64.abs().max()):.2e}, max rstd {float(rstd.max()):.1f} | library - fp64 stage: max {float(el.max()) / mx:.2e} rms {float(el.pow(2).mean().sqrt()) / mx:.2e}"
              f" | torch fp32 stage - fp64 stage: max {float(eo.max()) / mx:.2e} rms {float(eo.pow(2).mean().sqrt()) / mx:.2e}")
