"""Build-container only: the oracle against the REAL reference imported in place (/root/reference, stub recipe in refstub.py).
Skipped wherever the reference tree is absent (e.g. the GPU box) -- there the committed golden vectors pin the oracle."""
import pytest
import torch

import refstub

pytestmark = pytest.mark.skipif(not refstub.reference_available(), reason="/root/reference not present")


def test_forward_backward_match_reference_with_droppath_and_aux():
    from oracle import mome_oracle as O
    ref = refstub.load_reference()
    torch.manual_seed(0)
    M = ref.mome.ModalityAgnosticTransformer
    model = M(modalities=["img", None], num_classes=[5, None], tasks=["cls", None], embed_dim=32, depth=2, num_heads=2, vocab_size=50,
              max_text_len=8, drop_path_rate=0.0, with_aux=True, aux_trained=True)
    with torch.no_grad():
        for k, v in model.named_parameters():
            if "pos_embed" in k or "cls_token" in k:
                v.normal_(0, 0.02)
            if "cross_modal_scale" in k:
                v.fill_(0.3)
    model.train()
    img = torch.randn(3, 3, 224, 224) * 0.5
    y = torch.tensor([0, 3, 4])
    out = model([img, None])[0]
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    cfg = O.OracleCfg(modalities=("img", None), tasks=("cls", None), num_classes=(5, None), D=32, depth=2, heads=2, vocab=50,
                      max_text_len=8, with_aux=True, aux_trained=True)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    outs, cache = O.forward(p, cfg, [img, None])
    assert (outs[0] - out).abs().max() < 1e-5
    lo, dl = O.cross_entropy(outs[0], y)
    assert abs(float(lo) - float(loss)) < 1e-6
    g = O.backward(p, cfg, cache, [dl, None])
    for k, v in model.named_parameters():
        assert v.grad is not None, k
        scale = max(1e-6, float(v.grad.abs().max()))
        assert float((g[k] - v.grad).abs().max()) <= 2e-4 * scale + 1e-8, k


def test_adamw_step_matches_torch_optim():
    from oracle import mome_oracle as O
    torch.manual_seed(1)
    p = torch.randn(1000)
    g = torch.randn(1000) * 1e-2
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=0.0)
    m, v = torch.zeros(1000), torch.zeros(1000)
    q = p.clone()
    for step in (1, 2, 3):
        ref_p.grad = g.clone()
        opt.step()
        O.adamw_step(q, g, m, v, step, 1e-3)
    assert float((q - ref_p.detach()).abs().max()) <= 2.4e-7   # 1 ulp of values ~1.5


def test_scope_all_alias_keys_match_the_reference():
    """mome.py:824-827 on the real model: same state_dict key list (as a set and in the aliasing relation) as the flat-buffer mirror."""
    ref = refstub.load_reference()
    from fedcola_amd.mome import ModalityAgnosticTransformer as Mine
    kw = dict(modalities=["img", None], num_classes=[5, None], tasks=["cls", None], embed_dim=8, depth=2, num_heads=2, vocab_size=30, max_text_len=8)
    r = ref.mome.ModalityAgnosticTransformer(share_scope="all", **kw)
    r.sync_shared_weights()
    m = Mine(share_scope="all", **kw)
    m.sync_shared_weights()
    rk = [k for k in r.state_dict().keys() if not k.endswith("position_ids")]
    mk = list(m.state_dict().keys())
    assert set(rk) == set(mk), (set(rk) ^ set(mk))
    rsd, msd = r.state_dict(), m.state_dict()
    for k in rk:
        if k.startswith("blockses.1."):
            t = k.replace("blockses.1.", "blockses.0.", 1)
            assert rsd[k].data_ptr() == rsd[t].data_ptr() and msd[k].data_ptr() == msd[t].data_ptr()
    assert set(r.required_params().keys()) == set(m.required_params().keys())
    # colearn_param == 'blocks' changes nothing in the reference either
    kw2 = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=8, depth=2, num_heads=2, vocab_size=30, max_text_len=8)
    r2 = ref.mome.ModalityAgnosticTransformer(colearn_param="blocks", **kw2)
    r2.sync_shared_weights()
    sd2 = r2.state_dict()
    assert sd2["blockses.0.0.attn.qkv.weight"].data_ptr() != sd2["blockses.1.0.attn.qkv.weight"].data_ptr()
