"""Oracle (oracle/mome_oracle.py) against the golden vectors produced from the real reference."""
import pytest
import torch

import golden_util as G
from oracle import mome_oracle as O


def cfg_from_mk(mk):
    return O.OracleCfg(modalities=tuple(mk["modalities"]), tasks=tuple(mk["tasks"]), num_classes=tuple(mk["num_classes"]),
                       D=mk["embed_dim"], depth=mk["depth"], heads=mk["num_heads"], vocab=mk["vocab_size"],
                       max_text_len=mk["max_text_len"], with_aux=mk.get("with_aux", False),
                       aux_trained=mk.get("aux_trained", False), colearn_attn=mk.get("colearn_param", "none") == "attn")


@pytest.mark.parametrize("case", ["toy", "small", "imgcls_aux", "txtcls_aux", "colearn_attn"])
def test_model_case(case):
    rec = G.load(f"model_{case}.json")
    cfg = cfg_from_mk(rec["mk"])
    p = O.resolve_colearn(G.case_weights(case), cfg)
    img, ids, y = G.case_inputs(rec)
    kind = rec["kind"]
    batch = {"img+txt": ("img+txt", img, ids), "img": ("img", img, y), "txt": ("txt", ids, y)}[kind]
    state = dict(step=0, m={}, v={})
    loss, outs, grads = O.client_step(p, cfg, batch, state, lr=rec["lr"])
    assert abs(float(loss) - rec["loss"]) <= 2e-5 * max(1.0, abs(rec["loss"]))
    for o, r in zip(outs, rec["outs"]):
        G.compare(o, r, 1e-5, 1e-6, f"{case} outs")
    for k, r in rec["grads"].items():
        if r is None:
            assert k not in grads or float(grads[k].abs().max()) == 0.0, k
            continue
        G.compare(grads[k], r, 2e-4, 1e-7, f"{case} grad {k}")
    for k, r in rec["after"].items():
        G.compare_after_adamw(p[k], r, rec["grads"][k], rec["lr"], f"{case} after {k}")
    # eval-mode heads
    x = [img if kind != "txt" else None, ids if kind != "img" else None]
    ev, _ = O.forward(p, cfg, x, feat_out=False)      # golden eval pass ran after the optimizer step
    for o, r in zip(ev, rec["eval_outs"]):
        G.compare(o, r, 2e-5, 2e-6, f"{case} eval outs")


def test_backward_matches_autograd_fp64():
    """Self-check of the explicit backward against torch autograd in fp64 (incl. drop-path masks, aux)."""
    mk = dict(modalities=["img", None], num_classes=[7, None], tasks=["cls", None], embed_dim=16, depth=2,
              num_heads=2, vocab_size=20, max_text_len=8, with_aux=True, aux_trained=True)
    cfg = cfg_from_mk(mk)
    cfg.img_size, cfg.patch = 32, 16
    shapes = {"embeddings.0.pos_embed": (1, 5, 16), "embeddings.0.cls_token": (1, 1, 16),
              "embeddings.0.embed.proj.weight": (16, 3, 16, 16), "embeddings.0.embed.proj.bias": (16,),
              "norm.weight": (16,), "norm.bias": (16,), "heads.0.head.weight": (7, 16), "heads.0.head.bias": (7,)}
    for l in range(2):
        pre = f"blockses.0.{l}"
        for nm, shp in [("norm1", None), ("attn.qkv", (48, 16)), ("attn.proj", (16, 16)), ("norm2", None),
                        ("mlp.fc1", (64, 16)), ("mlp.fc2", (16, 64))]:
            if shp is None:
                shapes[f"{pre}.{nm}.weight"] = (16,)
                shapes[f"{pre}.{nm}.bias"] = (16,)
            else:
                shapes[f"{pre}.{nm}.weight"] = shp
                shapes[f"{pre}.{nm}.bias"] = (shp[0],)
                shapes[f"{pre}.{nm}.aux_weight"] = shp
                shapes[f"{pre}.{nm}.cross_modal_scale"] = (1,)
    from synth import det_state_dict, det_tensor
    p = {k: v.double().requires_grad_(True) for k, v in det_state_dict(shapes).items()}
    img = det_tensor((3, 3, 32, 32), 5, 0.5).double()
    y = torch.tensor([1, 4, 6])
    masks = {(0, 1, 0): torch.tensor([2.0, 0.0, 2.0], dtype=torch.float64), (0, 1, 1): torch.tensor([0.0, 2.0, 2.0], dtype=torch.float64)}
    outs, cache = O.forward(p, cfg, [img, None], dp_masks=masks)
    loss, dl = O.cross_entropy(outs[0], y)
    g = O.backward({k: v.detach() for k, v in p.items()}, cfg, cache_detach(cache), [dl.detach(), None])
    loss.backward()
    for k, v in p.items():
        assert v.grad is not None, k
        err = float((g[k] - v.grad).abs().max())
        assert err <= 1e-10 * max(1.0, float(v.grad.abs().max())), (k, err)


def cache_detach(c):
    if isinstance(c, torch.Tensor):
        return c.detach()
    if isinstance(c, dict):
        return {k: cache_detach(v) for k, v in c.items()}
    if isinstance(c, (list, tuple)):
        return type(c)(cache_detach(v) for v in c)
    return c
