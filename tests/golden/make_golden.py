#!/usr/bin/env python
"""Generate the committed golden vectors by running the REAL reference (/root/reference, imported in
place through tests/refstub.py).  Run in the build container only:

    python tests/golden/make_golden.py

Outputs (tests/golden/):
    model_<case>.json   forward outputs, loss, gradient / post-AdamW fingerprints (full tensors for the toy case)
    update_toy.json     FedavgClient.update() result dict + final weights fingerprint
    update_prox_toy.json  the same for FedproxClient.update() (mu = 0.5)
    update_clip_toy.json  the same for FedavgClient.update() with max_grad_norm = 1.0
    update_sgd_toy.json   the same with --optimizer SGD --momentum 0.9 --nesterov --weight_decay 1e-3
    server_update.json    three rounds of FedavgServer.update(): warm-up filter, freeze / unfreeze, aux refresh, LR decay
    agg.json            FedavgServer._aggregate outputs over the scope / compensation matrix
    agg_colearn.json    the same with colearn_param='attn' img+txt models (shared Attention tensors listed under both towers' keys)
    sampling.json       FedavgServer._sample_clients id lists
    init.json           default-init state_dict fingerprints under torch.manual_seed (factory order check)
    cream.json          CreamflClient.update() per modality and the server half of CreamflServer.update() on toy models
    split.json          simulate_split client -> index maps under np.random.seed (iid / unbalanced / caption sets / patho / diri)
    retrieval.json      COCOEvaluator.extract_features ordering, best ranks and recall scores on synthetic features

Inputs and weights are NOT stored: they come from tests/synth.py's exact integer generator, so a fixture
is only (generator seeds, expected outputs).
"""
from __future__ import annotations

import json
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from refstub import RefArgs, load_reference  # noqa: E402
from synth import det_ids, det_state_dict, det_tensor, summarize  # noqa: E402

ref = load_reference()
torch.set_num_threads(8)

# single-process restatement of torchmultimodal's ContrastiveLossWithTemperature (third party, absent):
import math  # noqa: E402


class ContrastiveLossWithTemperature(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.logit_scale = torch.nn.Parameter(torch.tensor(math.log(1 / 0.07)))

    def forward(self, a, b):
        self.logit_scale.data.clamp_(math.log(1.0), math.log(100.0))
        t = torch.exp(self.logit_scale)
        la = (a @ b.t()) * t
        lb = (b @ a.t()) * t
        lab = torch.arange(a.shape[0])
        return (torch.nn.functional.cross_entropy(la, lab) + torch.nn.functional.cross_entropy(lb, lab)) / 2


torch.nn.ContrastiveLoss = ContrastiveLossWithTemperature

CASES = {
    # name: (model kwargs, B, seq, kind)
    "toy": dict(mk=dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=4, depth=1,
                        num_heads=2, vocab_size=30, max_text_len=8), B=2, seq=8, kind="img+txt", full=True),
    "small": dict(mk=dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=128, depth=2,
                          num_heads=2, vocab_size=64, max_text_len=16), B=4, seq=16, kind="img+txt", full=False),
    "imgcls_aux": dict(mk=dict(modalities=["img", None], num_classes=[10, None], tasks=["cls", None], embed_dim=64, depth=2,
                               num_heads=1, vocab_size=64, max_text_len=16, with_aux=True, aux_trained=True),
                       B=4, seq=16, kind="img", full=False),
    "txtcls_aux": dict(mk=dict(modalities=[None, "txt"], num_classes=[None, 4], tasks=[None, "cls"], embed_dim=64, depth=2,
                               num_heads=1, vocab_size=64, max_text_len=16, with_aux=True, aux_trained=False),
                       B=4, seq=12, kind="txt", full=False),
    # colearn_param == 'attn' (mome.py:836-840): the text tower's Attention modules are the image tower's
    "colearn_attn": dict(mk=dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=64, depth=2,
                                 num_heads=1, vocab_size=64, max_text_len=16, colearn_param="attn"), B=4, seq=16, kind="img+txt", full=False),
}


def build(mk):
    m = ref.mome.ModalityAgnosticTransformer(**mk)
    m.sync_shared_weights()
    sd = m.state_dict()
    new = det_state_dict({k: tuple(v.shape) for k, v in sd.items()})
    m.load_state_dict(new, strict=True)
    return m


def pack(t, full):
    return dict(full=[float(x) for x in t.detach().reshape(-1)], shape=list(t.shape)) if full else summarize(t)


def model_case(name, c):
    m = build(c["mk"])
    m.train()
    B, seq, kind = c["B"], c["seq"], c["kind"]
    img = det_tensor((B, 3, 224, 224), 1000, 0.5)
    ids = det_ids((B, seq), 7, c["mk"]["vocab_size"])
    y = (torch.arange(B) * 3 + 1) % (max(x for x in c["mk"]["num_classes"] if x) if kind != "img+txt" else 1)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0)
    opt.zero_grad()
    if kind == "img+txt":
        outs = m([img, ids], feat_out=True)
        loss = torch.nn.ContrastiveLoss()(*outs)
    elif kind == "img":
        outs = m([img, None])
        loss = torch.nn.CrossEntropyLoss()(outs[0], y)
    else:
        outs = m([None, ids])
        loss = torch.nn.CrossEntropyLoss()(outs[1], y)
    loss.backward()
    rec = dict(case=name, mk=c["mk"], B=B, seq=seq, kind=kind, lr=1e-3, loss=float(loss),
               outs=[None if o is None else pack(o, True) for o in outs], grads={}, after={})
    for k, p in m.named_parameters():
        rec["grads"][k] = None if p.grad is None else pack(p.grad, c["full"])
    opt.step()
    for k, p in m.named_parameters():
        rec["after"][k] = pack(p, c["full"])
    # eval-mode forward through the heads (feat_out=False) as well
    m.eval()
    with torch.no_grad():
        ev = m([img if kind != "txt" else None, ids if kind != "img" else None], feat_out=False)
    rec["eval_outs"] = [None if o is None else pack(o, True) for o in ev]
    with open(os.path.join(HERE, f"model_{name}.json"), "w") as f:
        json.dump(rec, f)
    print(name, "loss", float(loss))


class SynthPairs(torch.utils.data.Dataset):
    def __init__(self, n, seq, vocab):
        self.img = det_tensor((n, 3, 224, 224), 2000, 0.5)
        self.ids = det_ids((n, seq), 11, vocab)

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.ids[i], i // 5, i, i


def update_case():
    """FedavgClient.update() on the toy img+txt model: 2 epochs x 3 batches (last one ragged)."""
    c = CASES["toy"]
    args = RefArgs(E=2, B=4, lr=1e-3, optimizer="AdamW", no_shuffle=True)
    ds = SynthPairs(10, 8, 30)
    cl = ref.fedavgclient.FedavgClient(args=args, training_set=ds, test_set=ds, task="rtv", modality="img+txt",
                                       eval_metrics=[], criterion="ContrastiveLoss")
    cl.id, cl.dataset, cl.device = 0, "Flickr30k", "cpu"
    cl.download({"Flickr30k": build(c["mk"])})
    res = cl.update()
    sd = cl.upload()
    rec = dict(results={str(k): v for k, v in res.items()}, n=10, B=4, E=2, lr=1e-3,
               after={k: pack(v, True) for k, v in sd.items() if v.dtype.is_floating_point})
    with open(os.path.join(HERE, "update_toy.json"), "w") as f:
        json.dump(rec, f)
    print("update", res)


def update_prox_case():
    """FedproxClient.update() (src/client/fedproxclient.py) on the toy img+txt model: 2 epochs x 3 batches, mu = 0.5."""
    c = CASES["toy"]
    args = RefArgs(E=2, B=4, lr=1e-3, optimizer="AdamW", no_shuffle=True, mu=0.5)
    ds = SynthPairs(10, 8, 30)
    cl = ref.fedproxclient.FedproxClient(args=args, training_set=ds, test_set=ds, task="rtv", modality="img+txt",
                                         eval_metrics=[], criterion="ContrastiveLoss")
    cl.id, cl.dataset, cl.device = 0, "Flickr30k", "cpu"
    cl.download({"Flickr30k": build(c["mk"])})
    res = cl.update()
    sd = cl.upload()
    rec = dict(results={str(k): v for k, v in res.items()}, n=10, B=4, E=2, lr=1e-3, mu=0.5,
               after={k: pack(v, True) for k, v in sd.items() if v.dtype.is_floating_point})
    with open(os.path.join(HERE, "update_prox_toy.json"), "w") as f:
        json.dump(rec, f)
    print("update prox", res)


def update_clip_case():
    """FedavgClient.update() with max_grad_norm = 1.0 (fedavgclient.py:98-99: clip_grad_norm_ between backward and the optimizer)."""
    c = CASES["toy"]
    args = RefArgs(E=2, B=4, lr=1e-3, optimizer="AdamW", no_shuffle=True, max_grad_norm=1.0)
    ds = SynthPairs(10, 8, 30)
    cl = ref.fedavgclient.FedavgClient(args=args, training_set=ds, test_set=ds, task="rtv", modality="img+txt",
                                       eval_metrics=[], criterion="ContrastiveLoss")
    cl.id, cl.dataset, cl.device = 0, "Flickr30k", "cpu"
    m = build(c["mk"])
    cl.download({"Flickr30k": m})
    # the clip must bite for the fixture to mean anything: record the first batch's total gradient norm
    mm = build(c["mk"])
    mm.train()
    img, ids = torch.stack([ds[i][0] for i in range(4)]), torch.stack([ds[i][1] for i in range(4)])
    torch.nn.ContrastiveLoss()(*mm([img, ids], feat_out=True)).backward()
    norm0 = float(torch.nn.utils.clip_grad_norm_(mm.parameters(), 1.0))
    assert norm0 > 1.5, norm0
    res = cl.update()
    sd = cl.upload()
    rec = dict(results={str(k): v for k, v in res.items()}, n=10, B=4, E=2, lr=1e-3, max_grad_norm=1.0, first_batch_grad_norm=norm0,
               after={k: pack(v, True) for k, v in sd.items() if v.dtype.is_floating_point})
    with open(os.path.join(HERE, "update_clip_toy.json"), "w") as f:
        json.dump(rec, f)
    print("update clip", res, "first-batch grad norm", norm0)


def update_sgd_case():
    """FedavgClient.update() with --optimizer SGD --momentum 0.9 --nesterov (main.py:269-273: SGD is the argument's default; _refine_optim_args,
    fedavgclient.py:34-42, hands torch.optim.SGD lr / momentum / weight_decay / nesterov from args)."""
    c = CASES["toy"]
    args = RefArgs(E=2, B=4, lr=1e-2, optimizer="SGD", no_shuffle=True, momentum=0.9, nesterov=True, weight_decay=1e-3)
    ds = SynthPairs(10, 8, 30)
    cl = ref.fedavgclient.FedavgClient(args=args, training_set=ds, test_set=ds, task="rtv", modality="img+txt",
                                       eval_metrics=[], criterion="ContrastiveLoss")
    cl.id, cl.dataset, cl.device = 0, "Flickr30k", "cpu"
    cl.download({"Flickr30k": build(c["mk"])})
    res = cl.update()
    sd = cl.upload()
    rec = dict(results={str(k): v for k, v in res.items()}, n=10, B=4, E=2, lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3,
               after={k: pack(v, True) for k, v in sd.items() if v.dtype.is_floating_point})
    with open(os.path.join(HERE, "update_sgd_toy.json"), "w") as f:
        json.dump(rec, f)
    print("update sgd", res)


def server_update_case():
    """The REAL FedavgServer.update() (fedavgserver.py:784-856) for three rounds on toy models with real FedavgClients: sampling with the
    warm-up filter (:307-308), the request fan-out with freeze / unfreeze (:505-520, 417-429), per-dataset _aggregate, the aux refresh
    (:821-845), LR decay (:851-852).  Recorded per round: sampled ids, the lr every trained client saw, requires_grad of every client
    parameter at the time of its update(), the clients' result dicts, curr_lr afterwards and every global model's full state_dict."""
    from unittest import mock
    from collections import defaultdict
    import fl_util as F
    args = RefArgs(**F.ROUND_ARGS)
    srv = object.__new__(ref.fedavgserver.FedavgServer)
    srv.args = args
    srv._round = 0
    srv.writer = mock.MagicMock()
    srv.results = defaultdict(dict)
    srv.curr_lr = args.lr
    srv.Cs = dict(F.ROUND_CS)
    srv.global_models = {}
    for i, ds in enumerate(F.ROUND_DS):
        m = ref.mome.ModalityAgnosticTransformer(with_aux=True, aux_trained=False, **F.round_model_kwargs(ds))
        m.sync_shared_weights()
        m.load_state_dict(det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, base_seed=31 * (i + 1)))
        srv.global_models[ds] = m
    srv._init_param_scope(args.shared_param, args.share_scope)
    clients = []
    for cid, ds, n in F.ROUND_LAYOUT:
        task, mod = F.ROUND_DS[ds]
        d = F.round_dataset(cid, ds, n)
        cl = ref.fedavgclient.FedavgClient(args=args, training_set=d, test_set=d, task=task, modality=mod,
                                           eval_metrics=["acc1"] if task == "cls" else [], criterion="CrossEntropyLoss" if task == "cls" else "ContrastiveLoss")
        cl.id, cl.dataset, cl.device = cid, ds, "cpu"
        clients.append(cl)
    srv._clients = clients
    trace = {}
    for cl in clients:
        def spy(cl=cl, orig=cl.update):
            trace[cl.id] = dict(lr=float(cl.args.lr), requires_grad={k: bool(p.requires_grad) for k, p in cl.model.named_parameters()})
            res = orig()
            trace[cl.id]["result"] = {str(k): v for k, v in res.items()}
            return res
        cl.update = spy
    random.seed(F.ROUND_SEED)
    rec = dict(scope=dict(srv.param_scope), rounds=[],
               init={ds: {k: pack(v, True) for k, v in m.state_dict().items() if v.dtype.is_floating_point} for ds, m in srv.global_models.items()})
    for r in range(1, F.ROUND_N + 1):
        srv._round = r
        trace.clear()
        ids = srv.update()
        assert all(c.model is None for c in clients)
        rec["rounds"].append(dict(round=r, ids=[int(i) for i in ids], curr_lr=float(srv.curr_lr),
                                  clients={str(k): v for k, v in trace.items()},
                                  models={ds: {k: pack(v, True) for k, v in m.state_dict().items() if v.dtype.is_floating_point}
                                          for ds, m in srv.global_models.items()}))
        print("server round", r, ids, srv.curr_lr, {k: sum(v["requires_grad"].values()) for k, v in trace.items()})
    # a fourth round with --fedavg_eval (fedavgserver.py:796-808): the models handed to _central_evaluate(fedavg=True) are what plain FedAvg would give;
    # the round then continues from the old models
    args.fedavg_eval = True
    seen = {}
    srv._central_evaluate = lambda fedavg=False: seen.update(
        fedavg=bool(fedavg), models={ds: {k: pack(v, True) for k, v in m.state_dict().items() if v.dtype.is_floating_point} for ds, m in srv.global_models.items()})
    srv._round = F.ROUND_N + 1
    trace.clear()
    ids = srv.update()
    assert seen["fedavg"] is True
    rec["fedavg_eval_round"] = dict(round=F.ROUND_N + 1, ids=[int(i) for i in ids], curr_lr=float(srv.curr_lr), evaluated=seen["models"],
                                    models={ds: {k: pack(v, True) for k, v in m.state_dict().items() if v.dtype.is_floating_point}
                                            for ds, m in srv.global_models.items()})
    args.fedavg_eval = False
    print("server round", F.ROUND_N + 1, "(fedavg_eval)", ids)
    # the three behaviours the fixture exists for must actually occur in it
    r1, r2, r3 = rec["rounds"]
    assert all(F.ROUND_DS[F.ROUND_LAYOUT[i][1]][1] == "txt" for i in r1["ids"])                                   # warm-up filter
    assert any(not v for c in r2["clients"].values() for k, v in c["requires_grad"].items() if "aux_weight" not in k)   # freeze
    assert any(v for i, c in r3["clients"].items() if F.ROUND_LAYOUT[int(i)][1] == "CIFAR100" for k, v in c["requires_grad"].items() if "aux_weight" in k)  # unfreeze quirk
    with open(os.path.join(HERE, "server_update.json"), "w") as f:
        json.dump(rec, f)
    print("server update ok")


# ----------------------------------------------------------------------------------------- aggregation
DS = {"CIFAR100": ("cls", "img"), "AG_NEWS": ("cls", "txt"), "Flickr30k": ("rtv", "img+txt")}


def agg_models(with_aux, colearn=None):
    out = {}
    common = dict(embed_dim=4, depth=1, num_heads=2, vocab_size=30, max_text_len=8)
    mm = dict(colearn_param=colearn) if colearn else {}
    out["CIFAR100"] = ref.mome.ModalityAgnosticTransformer(modalities=["img", None], num_classes=[100, None], tasks=["cls", None],
                                                           with_aux=with_aux, aux_trained=True, **common)
    out["AG_NEWS"] = ref.mome.ModalityAgnosticTransformer(modalities=[None, "txt"], num_classes=[None, 4], tasks=[None, "cls"],
                                                          with_aux=with_aux, aux_trained=True, **common)
    out["Flickr30k"] = ref.mome.ModalityAgnosticTransformer(modalities=["img", "txt"], num_classes=[None, None],
                                                            tasks=["rtv", "rtv"], with_aux=with_aux, aux_trained=True, **common, **mm)
    if colearn:
        out["Flickr30k"].sync_shared_weights()      # what every factory does after the constructor (mome.py:950,971,992,1031)
        assert out["Flickr30k"].blockses[1][0].attn is out["Flickr30k"].blockses[0][0].attn
    for i, (k, m) in enumerate(out.items()):
        sd = m.state_dict()
        m.load_state_dict(det_state_dict({kk: tuple(v.shape) for kk, v in sd.items()}, base_seed=31 * (i + 1)))
    return out


class FakeClient:
    def __init__(self, cid, dataset, model, args, n):
        self.id, self.dataset, self.args = cid, dataset, args
        self.task, self.modality = DS[dataset]
        self.model = model
        self.n = n
    upload = ref.fedavgclient.FedavgClient.upload

    def __len__(self):
        return self.n


def agg_case(colearn=None):
    """colearn='attn': the img+txt models are built with colearn_param='attn' (mome.py:836-840), so their state_dict lists the shared
    Attention tensors under both towers' keys and _aggregate's in-place blend visits the shared tensor twice -> agg_colearn.json."""
    recs = []
    # clients: 2 img, 2 txt, 2 img+txt ; sizes differ
    layout = [(0, "CIFAR100", 50), (1, "CIFAR100", 70), (2, "AG_NEWS", 40), (3, "AG_NEWS", 90), (4, "Flickr30k", 60), (5, "Flickr30k", 30)]
    combos = [("none", "dataset", False, False, [1, 1, 1]), ("attn", "modality", False, False, [1, 1, 1]),
              ("attn", "modality", True, False, [1, 1, 1]), ("blocks", "all", False, False, [1, 1, 1]),
              ("blocks", "all", True, False, [1, 1, 0.5]), ("attn", "modality", True, True, [1, 1, 1]),
              ("blocks", "modality", False, False, [1, 0.5, 1])]
    if colearn:
        combos = [("attn", "modality", False, False, [1, 1, 1]), ("attn", "modality", True, False, [1, 1, 0.5]), ("none", "dataset", False, False, [1, 1, 1])]
    for shared_param, share_scope, comp, with_aux, oms in combos:
        args = RefArgs(shared_param=shared_param, share_scope=share_scope, compensation=comp, with_aux=with_aux,
                       aux_trained=True, datasets=list(DS.keys()), modalities=["img", "txt", "img+txt"], out_modality_scales=oms)
        srv = object.__new__(ref.fedavgserver.FedavgServer)
        srv.args = args
        srv._round = 0
        srv.global_models = agg_models(with_aux, colearn)
        srv._init_param_scope(shared_param, share_scope)
        clients = []
        for cid, ds, n in layout:
            import copy
            m = copy.deepcopy(srv.global_models[ds])
            sd = m.state_dict()
            m.load_state_dict(det_state_dict({k: tuple(v.shape) for k, v in sd.items()}, base_seed=1000 + 17 * cid))
            clients.append(FakeClient(cid, ds, m, args, n))
        srv._clients = clients
        ids = [0, 1, 3, 4, 5] if shared_param != "none" else [0, 1, 2, 3, 4, 5]
        sizes = {i: clients[i].n for i in ids}
        rec = dict(shared_param=shared_param, share_scope=share_scope, compensation=comp, with_aux=with_aux, colearn=colearn,
                   out_modality_scales=oms, ids=ids, sizes={str(k): v for k, v in sizes.items()},
                   layout=layout, scope=dict(srv.param_scope), result={})
        for i, ds in enumerate(srv.global_models.keys()):
            srv.global_model = srv.global_models[ds]
            srv.task, srv.modality = DS[ds]
            srv.dataset = ds
            srv.out_modality_scale = oms[i]
            srv._aggregate(ids, sizes)
            rec["result"][ds] = {k: pack(v, True) for k, v in srv.global_model.state_dict().items() if v.dtype.is_floating_point}
        recs.append(rec)
        print("agg", shared_param, share_scope, comp, with_aux)
    with open(os.path.join(HERE, "agg_colearn.json" if colearn else "agg.json"), "w") as f:
        json.dump(recs, f)


def sampling_case():
    recs = []
    for seed in (1, 2, 7):
        for equal in (True, False):
            args = RefArgs(equal_sampled=equal, C=0.25, K=32, datasets=["CIFAR100", "AG_NEWS", "Flickr30k"])
            srv = object.__new__(ref.fedavgserver.FedavgServer)
            srv.args = args
            srv._round = 0
            srv.Cs = {"CIFAR100": 0.25, "AG_NEWS": 0.25, "Flickr30k": 0.25}
            cl = []
            for i in range(32):
                o = type("C", (), {})()
                o.id = i
                o.dataset = "CIFAR100" if i < 12 else ("AG_NEWS" if i < 24 else "Flickr30k")
                o.modality = "x"
                o.device = None
                cl.append(o)
            srv._clients = cl
            random.seed(seed)
            rounds = [srv._sample_clients() for _ in range(4)]
            recs.append(dict(seed=seed, equal_sampled=equal, rounds=rounds))
    with open(os.path.join(HERE, "sampling.json"), "w") as f:
        json.dump(recs, f)
    print("sampling ok")


def init_case():
    recs = []
    for name, mk in [("toy", CASES["toy"]["mk"]), ("small", CASES["small"]["mk"]), ("imgcls_aux", CASES["imgcls_aux"]["mk"]),
                     ("txtcls_aux", CASES["txtcls_aux"]["mk"]), ("colearn_attn", CASES["colearn_attn"]["mk"])]:
        torch.manual_seed(1234)
        m = ref.mome.ModalityAgnosticTransformer(**mk)
        m.sync_shared_weights()
        recs.append(dict(name=name, mk=mk, seed=1234,
                         keys=[[k, list(v.shape)] for k, v in m.state_dict().items()],
                         params=[k for k, _ in m.named_parameters()],
                         requires_grad={k: bool(p.requires_grad) for k, p in m.named_parameters()},
                         required=list(m.required_params().keys()),
                         aux=list(m.aux_params().keys()) if m.with_aux else None,
                         sd={k: summarize(v.float()) for k, v in m.state_dict().items()}))
    with open(os.path.join(HERE, "init.json"), "w") as f:
        json.dump(recs, f)
    print("init ok")


def cream_case():
    """CreamFL (src/client/creamflclient.py, src/server/creamflserver.py): CreamflClient.update() per modality (local epoch +
    public-set contrastive distillation with clip_grad_norm 2) and CreamflServer.update()'s server half (contrastive feature
    aggregation, zero-init weighted aggregate, KD distillation) on toy models."""
    import types
    import cream_util as CU
    from synth import det_state_dict as dsd

    def mk_model(kind):
        m = ref.mome.ModalityAgnosticTransformer(**CU.MK[kind])
        m.sync_shared_weights()
        m.load_state_dict(dsd({k: tuple(v.shape) for k, v in m.state_dict().items()}, base_seed=CU.BASE_SEED[kind]), strict=True)
        return m
    gi, gt = CU.global_features()
    pub = CU.PubSet()
    dindex = [pub.index[i] for i in range(len(pub))]
    rec = dict(clients={}, server={}, shapes={k: {n: list(v.shape) for n, v in mk_model(k).state_dict().items()} for k in CU.MK})
    clients = {}
    for cid, (kind, ds, task, crit, dsname) in enumerate([("img", CU.Cls("img"), "cls", "CrossEntropyLoss", "CIFAR100"),
                                                          ("txt", CU.Cls("txt", classes=4), "cls", "CrossEntropyLoss", "AG_NEWS"),
                                                          ("mm", CU.Pairs(), "rtv", "ContrastiveLoss", "Flickr30k")]):
        args = RefArgs(**CU.CREAM_ARGS)
        modality = {"img": "img", "txt": "txt", "mm": "img+txt"}[kind]
        cl = ref.creamflclient.CreamflClient(args=args, training_set=ds, test_set=ds, task=task, modality=modality,
                                             eval_metrics=["acc1"] if task == "cls" else [], criterion=crit)
        cl.id, cl.dataset, cl.device = cid, dsname, "cpu"
        cl.pub_dataset = pub
        cl.global_img_feature, cl.global_txt_feature, cl.distill_index = gi.clone(), gt.clone(), list(dindex)
        cl.download({dsname: mk_model(kind)})
        res = cl.update()
        r = dict(results={str(k): v for k, v in res.items()},
                 after={k: pack(v, False) for k, v in cl.model.state_dict().items() if v.dtype.is_floating_point})
        if kind != "mm":
            cl.update_pub_feature()
            r["pub_features"] = pack(cl.pub_features, True)
            r["distill_index"] = [int(x) for x in cl.distill_index]
        rec["clients"][kind] = r
        clients[cid] = cl
        print("cream client", kind, res)
    # ---- server half: the REAL CreamflServer.update() body on a shell (client update / sampling stubbed out)
    srv = object.__new__(ref.creamflserver.CreamflServer)
    srv._round, srv._clients = 1, [clients[0], clients[1], clients[2]]
    srv.args = RefArgs(datasets=["CIFAR100", "AG_NEWS", "Flickr30k"], modalities=["img", "txt", "img+txt"], lr_decay=0.99, lr_decay_step=1,
                       **CU.CREAM_ARGS)
    srv.global_models = {"CIFAR100": mk_model("img"), "AG_NEWS": mk_model("txt"), "Flickr30k": mk_model("mm")}
    srv._init_param_scope("none", "dataset")
    srv.device, srv.curr_lr = "cpu", 1e-3
    srv.global_img_feature, srv.global_txt_feature, srv.distill_index = gi.clone(), gt.clone(), list(dindex)
    srv.pub_loader = list(torch.utils.data.DataLoader(pub, batch_size=4, shuffle=False))
    clients[0].pub_features = CU.client_pub_features(81)
    clients[1].pub_features = CU.client_pub_features(82)
    extra_img = types.SimpleNamespace(id=3, modality="img", dataset="CIFAR100", task="cls", pub_features=CU.client_pub_features(83),
                                      upload=lambda: clients[0].upload(), training_set=list(range(9)))
    srv._clients.append(extra_img)
    sizes = {0: 6, 1: 6, 2: 6, 3: 9}
    srv._generate_public_logit = lambda: None
    srv._sample_clients = lambda: [0, 1, 2, 3]
    srv._request = lambda *a, **k: dict(sizes)
    srv._empty_client_models = lambda: None
    ids = srv.update()
    rec["server"] = dict(ids=ids, sizes={str(k): v for k, v in sizes.items()}, img_vec=pack(srv.img_vec, True), txt_vec=pack(srv.txt_vec, True),
                         curr_lr=srv.curr_lr,
                         after={ds: {k: pack(v, False) for k, v in m.state_dict().items() if v.dtype.is_floating_point}
                                for ds, m in srv.global_models.items()})
    with open(os.path.join(HERE, "cream.json"), "w") as f:
        json.dump(rec, f)
    print("cream ok")


SPLIT_CASES = [
    dict(name="iid", split_type="iid", dataset="CIFAR100", K=7, n=103, seed=3),
    dict(name="unbalanced", split_type="unbalanced", dataset="CIFAR100", K=5, n=211, seed=4),
    dict(name="flickr", split_type="unbalanced", dataset="Flickr30k", K=6, n=5 * 157, seed=5),
    dict(name="coco_noniid_name", split_type="diri", dataset="Coco", K=4, n=5 * 90, seed=6),     # caption sets ignore the type
    dict(name="patho", split_type="patho", dataset="CIFAR100", K=10, n=400, seed=7, mincls=2, num_classes=10),
    dict(name="diri", split_type="diri", dataset="CIFAR100", K=5, n=600, seed=8, cncntrtn=0.5, num_classes=6),
]


def split_case():
    """simulate_split (src/loaders/split.py) under np.random.seed: the client -> sample-index maps."""
    import importlib
    import numpy as np
    sp = importlib.import_module("src.loaders.split")
    recs = []
    for c in SPLIT_CASES:
        class DS:
            def __init__(self, n, ncls):
                self.targets = [(i * 7 + i // 3) % ncls for i in range(n)]

            def __len__(self):
                return len(self.targets)
        a = RefArgs(**{k: v for k, v in c.items() if k not in ("name", "n", "seed")})
        np.random.seed(c["seed"])
        m = sp.simulate_split(a, DS(c["n"], c.get("num_classes", 10)))
        tail = float(np.random.uniform())                       # the RNG stream position after the call is part of the contract
        recs.append(dict(cfg=c, map={str(k): [int(x) for x in v] for k, v in m.items()}, next_uniform=tail))
    with open(os.path.join(HERE, "split.json"), "w") as f:
        json.dump(recs, f)
    print("split ok", [(r["cfg"]["name"], [len(v) for v in r["map"].values()]) for r in recs])


def retrieval_case():
    """COCOEvaluator (src/metrics/eval_coco.py) on synthetic features: collected ordering, best ranks, scores."""
    import importlib
    import numpy as np
    from retrieval_util import RETRIEVAL_CASES, FakeDataset, FakeLoader, PassThroughModel, retrieval_set, stream
    ec = importlib.import_module("src.metrics.eval_coco")
    recs = {}
    for name, c in RETRIEVAL_CASES.items():
        img, cap, iids, aids = retrieval_set(c["n_images"], c["caps"], c["D"], c["seed"])
        batches = stream(img, cap, iids, aids, c["caps"], c["batch"])
        ev = ec.COCOEvaluator("matmul", n_crossfolds=c["folds"], extract_device="cpu", eval_device="cpu", verbose=False)
        ev.set_model(PassThroughModel(c["D"]))
        loader = FakeLoader(batches, FakeDataset(c["n_images"], cap.shape[0]))
        ex = ev.extract_features(loader)
        captured = []
        orig = ec.recall_at_k

        def spy(ranks, k):
            if k == 1:
                captured.append([int(r) for r in ranks])
            return orig(ranks, k)
        ec.recall_at_k = spy
        try:
            scores = ev.evaluate(loader, n_images_per_crossfold=c["ipf"], n_captions_per_crossfold=c["cpf"], eval_batch_size=64)
        finally:
            ec.recall_at_k = orig
        # evaluate() order: per fold (i2t, t2i), then full i2t, full t2i
        recs[name] = dict(cfg=c, image_ids=[int(v) for v in ex["image_ids"]], caption_ids=[int(v) for v in ex["caption_ids"]],
                          caption_classes=[int(v) for v in ex["caption_classes"]],
                          image_feature_sum=float(ex["image_features"].sum()), caption_feature_sum=float(ex["caption_features"].sum()),
                          ranks_i2t=captured[-2], ranks_t2i=captured[-1], fold_ranks=captured[:-2],
                          scores=json.loads(json.dumps(scores, default=float)))
    with open(os.path.join(HERE, "retrieval.json"), "w") as f:
        json.dump(recs, f)
    print("retrieval ok")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "retrieval":
        retrieval_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cream":
        cream_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "split":
        split_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "colearn":
        model_case("colearn_attn", CASES["colearn_attn"])
        init_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "agg_colearn":
        agg_case("attn")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "server_update":
        server_update_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sgd":
        update_sgd_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "clip":
        update_clip_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "prox":
        update_prox_case()
        sys.exit(0)
    for n, c in CASES.items():
        model_case(n, c)
    update_case()
    update_prox_case()
    update_clip_case()
    update_sgd_case()
    server_update_case()
    agg_case()
    agg_case("attn")
    sampling_case()
    init_case()
    retrieval_case()
    split_case()
    cream_case()
