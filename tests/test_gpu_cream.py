"""N2 (CreamFL) on the GPU: the loss / clipping / per-parameter-step optimizer kernels against the oracle, and the client / server
mirrors end to end against golden vectors produced by the reference's CreamflClient.update() / CreamflServer.update()."""
import types

import numpy as np
import pytest
import torch

import cream_util as CU
import golden_util as G
from oracle import creamfl_oracle as CO
from refstub import RefArgs
from synth import det_state_dict

pytestmark = pytest.mark.gpu
GOLD = G.load("cream.json")


def lib():
    from fedcola_amd import _lib
    return _lib


def product_model(kind):
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    m = M(precision="fp32", init=False, **CU.MK[kind])
    m.load_state_dict(det_state_dict({k: tuple(v) for k, v in GOLD["shapes"][kind].items()}, base_seed=CU.BASE_SEED[kind]))
    return m.cuda()


def check_after(sd, rec_after, tol, what):
    for k, r in rec_after.items():
        if k.endswith("attn.qkv.bias"):
            continue
        G.compare(sd[k].detach().cpu(), r, tol, tol, f"{what} {k}")


def test_loss_kernels_vs_oracle():
    _lib = lib()
    L, P_ = _lib.lib(), _lib.ptr
    gen = torch.Generator().manual_seed(4)
    B, D, P = 7, 24, 37
    f = torch.nn.functional.normalize(torch.randn(B, D, generator=gen), dim=-1)
    t, o = torch.randn(B, D, generator=gen), torch.randn(B, D, generator=gen)
    Gm = torch.randn(P, D, generator=gen)
    lab = torch.randint(0, P, (B,), generator=gen)
    fd, td, od, Gd, labd = f.cuda(), t.cuda(), o.cuda(), Gm.cuda(), lab.cuda()
    lossbuf = torch.zeros(2, device="cuda")
    df = torch.full((B, D), float("nan"), device="cuda")
    sp = _lib.stream_ptr()
    w = 0.5
    _lib.check(L.fc_cream_moon_loss(P_(fd), P_(td), P_(od), B, D, 2 * B, w, P_(lossbuf), P_(df), 0, sp))
    scratch = torch.empty(L.fc_cream_inter_scratch_floats(B, P), device="cuda")
    _lib.check(L.fc_cream_inter_loss(P_(fd), P_(Gd), P_(labd), B, P, D, w, P_(scratch), scratch.numel(), P_(lossbuf), P_(df), 1, sp))
    torch.cuda.synchronize()
    l1, d1 = CO.moon_ce(f, t, o, rows_norm=2 * B)
    l2, d2 = CO.inter_ce(f, Gm, lab)
    assert float(lossbuf[1]) == pytest.approx(w * float(l1 + l2), rel=1e-5)
    assert float(lossbuf[0]) == pytest.approx(w * float(l1 + l2) * B, rel=1e-5)
    assert (df.cpu() - w * (d1 + d2)).abs().max() <= 1e-6
    # gather + MSE
    idx = torch.tensor([3, 0, 36, 5, 5, 1, 2]).cuda()
    g = torch.empty(B, D, device="cuda")
    _lib.check(L.fc_gather_rows(P_(Gd), P_(idx), B, D, P_(g), sp))
    assert torch.equal(g.cpu(), Gm[idx.cpu()])
    lossbuf.zero_()
    dout = torch.empty(B, D, device="cuda")
    _lib.check(L.fc_mse_loss_fwd_bwd(P_(fd), P_(g), B * D, 0.3, B, P_(lossbuf), P_(dout), sp))
    ref = 0.3 * torch.nn.functional.mse_loss(f, Gm[idx.cpu()])
    assert float(lossbuf[1]) == pytest.approx(float(ref), rel=1e-5)
    assert (dout.cpu() - 0.3 * 2 * (f - Gm[idx.cpu()]) / (B * D)).abs().max() <= 1e-7
    # server aggregation
    vecs = [torch.nn.functional.normalize(torch.randn(P, D, generator=gen), dim=-1) for _ in range(3)]
    Gn = torch.nn.functional.normalize(Gm, dim=-1)
    from fedcola_amd.server.creamflserver import CreamflServer
    srv = object.__new__(CreamflServer)
    srv.device = "cuda"
    out = srv.aggregate_features([v.cuda() for v in vecs], Gn.cuda())
    assert (out.cpu() - CO.aggregate_features(vecs, Gn)).abs().max() <= 1e-6
    assert srv.aggregate_features([], Gn.cuda()) is None


def test_clip_and_per_parameter_adam_vs_torch():
    """fc_clip_grad_norm and fc_adamw_step_segs against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW with a parameter
    whose gradient is None in the second step (skipped, step count not advanced)."""
    _lib = lib()
    L, P_ = _lib.lib(), _lib.ptr
    m = product_model("img")
    segs = list(m.segments.items())
    n = m.flat.numel()
    gen = torch.Generator().manual_seed(1)
    p0 = m.flat.detach().clone()
    params = {k: torch.nn.Parameter(p0[s["offset"]: s["offset"] + s["numel"]].cpu().clone()) for k, s in segs}
    opt = torch.optim.AdamW(params.values(), lr=1e-2, weight_decay=0.01)
    grads = torch.zeros(n, device="cuda")
    m1, m2 = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    scratch = torch.empty(L.fc_clip_scratch_bytes(m._handle.h), dtype=torch.uint8, device="cuda")
    norm_out = torch.zeros(1, device="cuda")
    seg_t = np.zeros(len(segs), dtype=np.int32)
    for it in range(3):
        mask = np.ones(len(segs), dtype=np.int32)
        if it == 1:
            mask[[i for i, (k, _) in enumerate(segs) if k.startswith("heads.")]] = 0
        grads.zero_()
        for i, (k, s) in enumerate(segs):
            g = torch.randn(s["numel"], generator=gen) * (5.0 if it == 0 else 0.01)      # step 0 clips, later steps do not
            if mask[i]:
                params[k].grad = g.clone()
                grads[s["offset"]: s["offset"] + s["numel"]] = g.cuda()
            else:
                params[k].grad = None
        tn = torch.nn.utils.clip_grad_norm_([q for q in params.values() if q.grad is not None], 2.0)
        opt.step()
        _lib.check(L.fc_clip_grad_norm(m._handle.h, P_(grads), 2.0, P_(scratch), scratch.numel(), P_(norm_out), _lib.stream_ptr()))
        seg_t += mask
        steps = np.ascontiguousarray(seg_t * mask, dtype=np.int32)
        _lib.check(L.fc_adamw_step_segs(m._handle.h, P_(m.flat), P_(grads), P_(m1), P_(m2), 1e-2, 0.9, 0.999, 1e-8, 0.01, steps.ctypes.data,
                                        len(steps), None, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert float(norm_out) == pytest.approx(float(tn), rel=1e-5)
        for k, s in segs:
            got = m.flat.detach()[s["offset"]: s["offset"] + s["numel"]].cpu()
            assert (got - params[k].detach()).abs().max() <= 2e-6 * max(1.0, float(params[k].abs().max())), (it, k)


@pytest.mark.parametrize("kind", ["img", "txt", "mm"])
def test_client_update_vs_reference_golden(kind):
    from fedcola_amd.client.creamflclient import CreamflClient
    rec = GOLD["clients"][kind]
    args = RefArgs(precision="fp32", **CU.CREAM_ARGS)
    modality = {"img": "img", "txt": "txt", "mm": "img+txt"}[kind]
    ds = CU.Pairs() if kind == "mm" else CU.Cls(kind, classes=10 if kind == "img" else 4)
    cl = CreamflClient(args=args, training_set=ds, test_set=ds, task="rtv" if kind == "mm" else "cls", modality=modality,
                       eval_metrics=[] if kind == "mm" else ["acc1"], criterion="ContrastiveLoss" if kind == "mm" else "CrossEntropyLoss")
    cl.id, cl.dataset, cl.device = 0, "x", "cuda"
    pub = CU.PubSet()
    cl.pub_dataset = pub
    gi, gt = CU.global_features()
    cl.global_img_feature, cl.global_txt_feature = gi.cuda(), gt.cuda()
    cl.distill_index = [pub.index[i] for i in range(len(pub))]
    cl.model = product_model(kind)
    res = cl.update()
    assert abs(res[1]["loss"] - rec["results"]["1"]["loss"]) <= 1e-4 * max(1.0, abs(res[1]["loss"]))
    if kind != "mm":
        assert res[1]["metrics"]["acc1"] == pytest.approx(rec["results"]["1"]["metrics"]["acc1"], abs=1e-12)
    check_after(cl.model.state_dict(), rec["after"], 3e-3, f"cream client {kind}")
    if kind != "mm":
        cl.update_pub_feature()
        assert [int(i) for i in cl.distill_index] == rec["distill_index"]
        G.compare(cl.pub_features.cpu(), rec["pub_features"], 5e-3, 5e-3, "pub features")


def test_server_half_vs_reference_golden():
    """The server half of CreamflServer.update() -- feature aggregation, zero-init aggregate + KD distillation of the img+txt model,
    fedavg=True aggregation of the uni-modal models -- with the client side stubbed exactly like the golden generator did."""
    from fedcola_amd.client.creamflclient import CreamflClient
    from fedcola_amd.server.creamflserver import CreamflServer
    rec = GOLD["server"]
    pub = CU.PubSet()
    gi, gt = CU.global_features()
    dindex = [pub.index[i] for i in range(len(pub))]
    clients = []
    for cid, (kind, ds, task, crit, dsname) in enumerate([("img", CU.Cls("img"), "cls", "CrossEntropyLoss", "CIFAR100"),
                                                          ("txt", CU.Cls("txt", classes=4), "cls", "CrossEntropyLoss", "AG_NEWS"),
                                                          ("mm", CU.Pairs(), "rtv", "ContrastiveLoss", "Flickr30k")]):
        args = RefArgs(precision="fp32", with_aux=False, **CU.CREAM_ARGS)
        cl = CreamflClient(args=args, training_set=ds, test_set=ds, task=task, modality={"img": "img", "txt": "txt", "mm": "img+txt"}[kind],
                           eval_metrics=["acc1"] if task == "cls" else [], criterion=crit)
        cl.id, cl.dataset, cl.device, cl.pub_dataset = cid, dsname, "cuda", pub
        cl.global_img_feature, cl.global_txt_feature, cl.distill_index = gi.cuda(), gt.cuda(), list(dindex)
        cl.model = product_model(kind)
        cl.update()                                  # the uploads are the post-update models, as in the golden run
        clients.append(cl)
    clients[0].pub_features, clients[1].pub_features = CU.client_pub_features(81).cuda(), CU.client_pub_features(82).cuda()
    extra = types.SimpleNamespace(id=3, modality="img", dataset="CIFAR100", task="cls", pub_features=CU.client_pub_features(83).cuda(),
                                  model=clients[0].model, upload=lambda: clients[0].upload(), training_set=list(range(9)))
    srv = object.__new__(CreamflServer)
    srv._round, srv._clients = 1, clients + [extra]
    srv.args = RefArgs(datasets=["CIFAR100", "AG_NEWS", "Flickr30k"], modalities=["img", "txt", "img+txt"], lr_decay=0.99, lr_decay_step=1,
                       precision="fp32", with_aux=False, **CU.CREAM_ARGS)
    srv.global_models = {"CIFAR100": product_model("img"), "AG_NEWS": product_model("txt"), "Flickr30k": product_model("mm")}
    srv._init_param_scope("none", "dataset")
    srv.device, srv.curr_lr = "cuda", 1e-3
    srv.results = {1: {}}
    srv.global_img_feature, srv.global_txt_feature, srv.distill_index = gi.cuda(), gt.cuda(), list(dindex)
    srv.pub_dataset = pub
    srv.pub_loader = torch.utils.data.DataLoader(pub, batch_size=4, shuffle=False)
    sizes = {int(k): v for k, v in rec["sizes"].items()}
    srv._generate_public_logit = lambda: None
    srv._sample_clients = lambda: [0, 1, 2, 3]
    srv._request = lambda *a, **k: dict(sizes)
    srv._empty_client_models = lambda: None
    ids = srv.update()
    assert ids == rec["ids"] and srv.curr_lr == pytest.approx(rec["curr_lr"])
    G.compare(srv.img_vec.cpu(), rec["img_vec"], 2e-5, 2e-6, "img_vec")
    G.compare(srv.txt_vec.cpu(), rec["txt_vec"], 2e-5, 2e-6, "txt_vec")
    for ds in ("Flickr30k", "CIFAR100", "AG_NEWS"):
        check_after(srv.global_models[ds].state_dict(), rec["after"][ds], 4e-3, f"server {ds}")


def test_public_logit_generation_matches_model_forward():
    from fedcola_amd.server.creamflserver import CreamflServer
    srv = object.__new__(CreamflServer)
    pub = CU.PubSet()
    srv.device = "cuda"
    srv.global_models = {"Flickr30k": product_model("mm")}
    srv.pub_loader = torch.utils.data.DataLoader(pub, batch_size=4, shuffle=False)
    srv._generate_public_logit()
    assert tuple(srv.global_img_feature.shape) == (CU.P, CU.D) and [int(i) for i in srv.distill_index] == [int(i) for i in pub.index]
    from oracle import mome_oracle as O
    from test_oracle_golden import cfg_from_mk
    p = det_state_dict({k: tuple(v) for k, v in GOLD["shapes"]["mm"].items()}, base_seed=CU.BASE_SEED["mm"])
    outs, _ = O.forward(p, cfg_from_mk(CU.MK["mm"]), [pub.img, pub.ids], feat_out=False)
    assert (srv.global_img_feature.cpu() - outs[0]).abs().max() <= 1e-4 and (srv.global_txt_feature.cpu() - outs[1]).abs().max() <= 1e-4


def test_two_epochs_uni_modal_uses_per_parameter_steps():
    """E = 2 on a uni-modal client: after the first distillation pass the head's Adam step count lags, so the second epoch's local
    steps go through the per-segment optimizer (not the fused global-step call).  Checked against the oracle (torch.optim semantics)."""
    from fedcola_amd.client.creamflclient import CreamflClient
    from test_oracle_golden import cfg_from_mk
    kind = "img"
    a = dict(CU.CREAM_ARGS)
    a["E"] = 2
    args = RefArgs(precision="fp32", **a)
    ds = CU.Cls("img")
    cl = CreamflClient(args=args, training_set=ds, test_set=ds, task="cls", modality="img", eval_metrics=["acc1"], criterion="CrossEntropyLoss")
    cl.id, cl.dataset, cl.device = 0, "CIFAR100", "cuda"
    pub = CU.PubSet()
    cl.pub_dataset = pub
    gi, gt = CU.global_features()
    cl.global_img_feature, cl.global_txt_feature = gi.cuda(), gt.cuda()
    cl.distill_index = [pub.index[i] for i in range(len(pub))]
    cl.model = product_model(kind)
    res = cl.update()
    p = det_state_dict({k: tuple(v) for k, v in GOLD["shapes"][kind].items()}, base_seed=CU.BASE_SEED[kind])
    pubs = [(pub.img[s:s + 4], pub.ids[s:s + 4], pub.index[s:s + 4]) for s in range(0, len(pub), 4)]
    tb = [("img", ds.x[s:s + 4], ds.y[s:s + 4]) for s in range(0, len(ds), 4)]
    exp = CO.client_update(p, cfg_from_mk(CU.MK[kind]), "img", tb, pubs, [int(i) for i in pub.index], gi, gt, E=2, lr=a["lr"],
                           interintra_weight=a["interintra_weight"], n_train=len(ds))
    for e in (1, 2):
        assert abs(res[e]["loss"] - exp[e]) <= 2e-4 * max(1.0, abs(exp[e])), e
    sd = cl.model.state_dict()
    for k, v in p.items():
        if not v.dtype.is_floating_point or k.endswith("attn.qkv.bias"):
            continue
        assert (sd[k].cpu() - v).abs().max() <= 6e-3, k
    # the head really lagged: a fused global-step run would have used step 6 for its second-epoch updates
    assert float((sd["heads.0.head.weight"].cpu() - p["heads.0.head.weight"]).abs().max()) <= 2e-3


def test_mm_client_at_vit_s_width_bf16_vs_emulating_oracle():
    """BASELINE.json config[4]'s client in the timed mode: an img+txt CreamFL client at the ViT-S width (384, 6 heads; 2 layers) in
    bf16 -- local contrastive epoch + public-set distillation (moon / inter losses on bf16 features, clip, AdamW) -- against the
    oracle emulating the kernels' bf16 rounding.  AdamW normalises the step, so the weights are compared through the size of the
    update (|dp| <= 1.5 lr per step) and through the agreement of its direction where the oracle's update is not marginal."""
    from fedcola_amd.client.creamflclient import CreamflClient
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    from oracle import mome_oracle as O
    from test_oracle_golden import cfg_from_mk
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=384, depth=2, num_heads=6,
              vocab_size=CU.VOCAB, max_text_len=CU.SEQ)
    a = dict(CU.CREAM_ARGS)
    args = RefArgs(precision="bf16", **a)
    ds, pub = CU.Pairs(), CU.PubSet()
    cl = CreamflClient(args=args, training_set=ds, test_set=ds, task="rtv", modality="img+txt", eval_metrics=[], criterion="ContrastiveLoss")
    cl.id, cl.dataset, cl.device = 0, "x", "cuda"
    cl.pub_dataset = pub
    g = torch.Generator().manual_seed(5)
    gi = torch.nn.functional.normalize(torch.randn(len(pub), 384, generator=g), dim=-1)
    gt = torch.nn.functional.normalize(torch.randn(len(pub), 384, generator=g), dim=-1)
    cl.global_img_feature, cl.global_txt_feature = gi.cuda(), gt.cuda()
    cl.distill_index = [pub.index[i] for i in range(len(pub))]
    torch.manual_seed(4)
    sd0 = {k: v.clone() for k, v in M(**mk).state_dict().items()}
    m = M(precision="bf16", init=False, **mk)
    m.load_state_dict(sd0)
    cl.model = m.cuda()
    res = cl.update()
    p = {k: v.clone() for k, v in sd0.items()}
    pubs = [(pub.img[s:s + 4], pub.ids[s:s + 4], pub.index[s:s + 4]) for s in range(0, len(pub), 4)]
    tb = [("img+txt", ds.img[s:s + 4], ds.ids[s:s + 4]) for s in range(0, len(ds), 4)]
    with O.emulate_bf16():
        exp = CO.client_update(p, cfg_from_mk(mk), "img+txt", tb, pubs, [int(i) for i in pub.index], gi, gt, E=1, lr=a["lr"],
                               interintra_weight=a["interintra_weight"], n_train=len(ds))
    assert abs(res[1]["loss"] - exp[1]) <= 3e-2 * max(1.0, abs(exp[1]))
    steps = len(tb) + len(pubs)
    sd = cl.model.state_dict()
    agree = total = 0
    for k, v in p.items():
        if not v.dtype.is_floating_point or k.endswith("attn.qkv.bias"):
            continue
        got, d_o = sd[k].cpu(), v - sd0[k]
        assert float((got - sd0[k]).abs().max()) <= 1.5 * a["lr"] * steps, k
        sel = d_o.abs() >= 0.5 * a["lr"] * steps                      # elements the oracle moved decisively
        agree += int((torch.sign(got - sd0[k])[sel] == torch.sign(d_o)[sel]).sum())
        total += int(sel.sum())
    assert total > 1000 and agree >= 0.97 * total, (agree, total)
