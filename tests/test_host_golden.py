"""CPU tests: the oracle and the product's HOST logic (sampling, scope table, coefficients, blend plan, C-ABI surface)
against the golden vectors generated from the reference."""
import ctypes
import os
import random
import re

import pytest
import torch

import golden_util as G
import host_util as H
from oracle import aggregate_oracle as AO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sampling_bit_exact_oracle_and_product():
    from fedcola_amd.server.fedavgserver import FedavgServer
    from refstub import RefArgs
    for rec in G.load("sampling.json"):
        datasets = ["CIFAR100", "AG_NEWS", "Flickr30k"]
        cds = ["CIFAR100" if i < 12 else ("AG_NEWS" if i < 24 else "Flickr30k") for i in range(32)]
        rng = random.Random(rec["seed"])
        got = [AO.sample_clients(rng, cds, datasets, {d: 0.25 for d in datasets}, rec["equal_sampled"], 0.25, 32) for _ in range(4)]
        assert got == rec["rounds"]
        srv = object.__new__(FedavgServer)
        srv.args = RefArgs(equal_sampled=rec["equal_sampled"], C=0.25, K=32, datasets=datasets)
        srv._round = 0
        srv.Cs = {d: 0.25 for d in datasets}
        cl = []
        for i in range(32):
            o = type("C", (), {})()
            o.id, o.dataset, o.modality = i, cds[i], "x"
            cl.append(o)
        srv._clients = cl
        random.seed(rec["seed"])
        assert [srv._sample_clients() for _ in range(4)] == rec["rounds"]


@pytest.mark.parametrize("idx", range(7))
def test_aggregation_oracle_vs_golden(idx):
    rec = G.load("agg.json")[idx]
    srv = H.make_server(rec)                                   # used here only as a container of models / client metadata
    assert AO.param_scope(list(rec["scope"].keys()), rec["shared_param"], rec["share_scope"]) == rec["scope"]
    assert srv.param_scope == rec["scope"]
    ids = rec["ids"]
    sizes = {i: len(srv.clients[i]) for i in ids}
    infos = {c.id: AO.ClientInfo(c.dataset, c.task, c.modality) for c in srv.clients}
    layers = ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")
    uploads = {}
    for i in ids:
        sd = {k: v.clone() for k, v in srv.clients[i].model.state_dict().items()}
        uploads[i] = AO.upload_fold(sd, layers) if (rec["with_aux"] and srv.clients[i].modality != "img+txt") else sd
    for n, ds in enumerate(srv.global_models):
        gm = srv.global_models[ds]
        g = {k: v.clone() for k, v in gm.required_params().items()}
        coef = AO.coefficients(list(g.keys()), rec["scope"], ids, sizes, infos, dataset=ds, task=H.DS[ds][0], modality=H.DS[ds][1],
                               out_modality_scale=rec["out_modality_scales"][n], compensation=rec["compensation"],
                               share_scope=rec["share_scope"], arg_modalities=["img", "txt", "img+txt"])
        out = AO.sequential_blend(g, uploads, ids, coef)
        for k, r in rec["result"][ds].items():
            if k in out:
                G.compare(out[k], r, 1e-6, 1e-7, f"oracle agg {ds} {k}")
        # closed form == sequential
        for k in out:
            part = [i for i in ids if k in uploads[i] and coef[k][i] != 0]
            wg, w = AO.effective_weights([coef[k][i] for i in part])
            cf = wg * g[k] + sum(wj * uploads[i][k] for wj, i in zip(w, part))
            assert (cf - out[k]).abs().max() <= 1e-6 * max(1.0, float(out[k].abs().max()))


@pytest.mark.parametrize("idx", range(7))
def test_product_host_plan_vs_golden(idx):
    """Product coefficient table + closed-form weights + blend plan (offsets across differently laid-out client buffers),
    executed with a CPU stand-in for the HIP blend kernel."""
    rec = G.load("agg.json")[idx]
    srv = H.make_server(rec)
    if rec["with_aux"]:
        # upload() folds on the device; on the CPU emulate the fold so that only the host logic is under test
        layers = ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")
        for c in srv.clients:
            if c.modality == "img+txt":
                continue
            def upload(c=c):
                m = c.model
                folded = m.flat.data.clone()
                sd = AO.upload_fold({k: v.clone() for k, v in m.state_dict().items()}, layers)
                for k, v in sd.items():
                    s = m.segments[k]
                    folded[s["offset"]: s["offset"] + s["numel"]] = v.reshape(-1)
                c._folded = folded
                return sd
            c.upload = upload
    H.run_aggregation(srv, rec, local_partial=H.cpu_local_partial)
    H.check_aggregation(srv, rec)


@pytest.mark.parametrize("idx", range(3))
def test_colearn_attn_aggregation_vs_golden(idx):
    """img+txt models built with colearn_param='attn' (mome.py:836-840): state_dict() lists the shared Attention tensors under both
    towers' keys and the reference's in-place blend visits the shared tensor once per key for every client (agg_colearn.json = the
    real FedavgServer._aggregate).  Both the oracle (shared tensors under both keys) and the product plan (one row per tensor, closed
    form over the interleaved sequence) must reproduce it."""
    rec = G.load("agg_colearn.json")[idx]
    srv = H.make_server(rec)
    gm = srv.global_models["Flickr30k"]
    assert any(k.startswith("blockses.1.") and ".attn." in k for k in gm.required_params()), "alias keys must be listed like the reference's"
    expect = H.oracle_sequential_blend(srv, rec)
    for ds, exp in rec["result"].items():
        for k, r in exp.items():
            if k in expect[ds]:
                G.compare(expect[ds][k], r, 1e-6, 1e-7, f"oracle colearn agg {ds} {k}")
    H.run_aggregation(srv, rec, local_partial=H.cpu_local_partial)
    H.check_aggregation(srv, rec)


def test_update_result_schema_oracle_vs_golden():
    """The oracle client loop (2 epochs x 3 batches, ragged last batch) reproduces FedavgClient.update()'s result dict."""
    from oracle import mome_oracle as O
    from synth import det_ids, det_tensor
    from test_oracle_golden import cfg_from_mk
    rec = G.load("update_toy.json")
    mk = G.load("model_toy.json")["mk"]
    cfg = cfg_from_mk(mk)
    p = G.case_weights("toy")
    img = det_tensor((rec["n"], 3, 224, 224), 2000, 0.5)
    ids = det_ids((rec["n"], 8), 11, 30)
    state = dict(step=0, m={}, v={})
    for e in range(rec["E"]):
        tot = 0.0
        for b0 in range(0, rec["n"], rec["B"]):
            sl = slice(b0, min(rec["n"], b0 + rec["B"]))
            loss, _, _ = O.client_step(p, cfg, ("img+txt", img[sl], ids[sl]), state, lr=rec["lr"])
            tot += float(loss) * (sl.stop - sl.start)
        assert abs(tot / rec["n"] - rec["results"][str(e + 1)]["loss"]) <= 2e-4
        assert rec["results"][str(e + 1)]["metrics"] == {}


def test_sgd_update_oracle_vs_golden():
    """--optimizer SGD --momentum 0.9 --nesterov --weight_decay 1e-3 (main.py:269-273): the oracle's torch.optim.SGD restatement in the client loop
    against the reference's FedavgClient.update() (update_sgd_toy.json): epoch losses and every weight."""
    from oracle import mome_oracle as O
    from synth import det_ids, det_tensor
    from test_oracle_golden import cfg_from_mk
    rec = G.load("update_sgd_toy.json")
    cfg = cfg_from_mk(G.load("model_toy.json")["mk"])
    p = G.case_weights("toy")
    img = det_tensor((rec["n"], 3, 224, 224), 2000, 0.5)
    ids = det_ids((rec["n"], 8), 11, 30)
    state = dict(step=0, m={}, v={})
    for e in range(rec["E"]):
        tot = 0.0
        for b0 in range(0, rec["n"], rec["B"]):
            sl = slice(b0, min(rec["n"], b0 + rec["B"]))
            loss, _, _ = O.client_step(p, cfg, ("img+txt", img[sl], ids[sl]), state, lr=rec["lr"], weight_decay=rec["weight_decay"],
                                       sgd=(rec["momentum"], rec["nesterov"]))
            tot += float(loss) * (sl.stop - sl.start)
        assert abs(tot / rec["n"] - rec["results"][str(e + 1)]["loss"]) <= 2e-4
    for k, r in rec["after"].items():
        exp = torch.tensor(r["full"]).reshape(r["shape"])
        assert float((p[k] - exp).abs().max()) <= 2e-5 * max(1.0, float(exp.abs().max())), k      # SGD is linear in the gradient: no Adam sign sensitivity


def test_fedprox_update_oracle_vs_golden():
    """N3: the oracle loop with the proximal term (un-squared per-tensor norms) reproduces FedproxClient.update()'s result dict
    and the post-update weights of the reference (tests/golden/update_prox_toy.json)."""
    from oracle import mome_oracle as O
    from synth import det_ids, det_tensor
    from test_oracle_golden import cfg_from_mk
    rec = G.load("update_prox_toy.json")
    cfg = cfg_from_mk(G.load("model_toy.json")["mk"])
    p = G.case_weights("toy")
    glob = {k: v.clone() for k, v in p.items()}
    img = det_tensor((rec["n"], 3, 224, 224), 2000, 0.5)
    ids = det_ids((rec["n"], 8), 11, 30)
    state = dict(step=0, m={}, v={})
    for e in range(rec["E"]):
        tot = 0.0
        for b0 in range(0, rec["n"], rec["B"]):
            sl = slice(b0, min(rec["n"], b0 + rec["B"]))
            loss, _, _ = O.client_step(p, cfg, ("img+txt", img[sl], ids[sl]), state, lr=rec["lr"], prox=(glob, rec["mu"]))
            tot += float(loss) * (sl.stop - sl.start)
        assert abs(tot / rec["n"] - rec["results"][str(e + 1)]["loss"]) <= 2e-4
    # the proximal term changes the trajectory: the plain-FedAvg golden losses differ by > 1e-2
    assert abs(rec["results"]["2"]["loss"] - G.load("update_toy.json")["results"]["2"]["loss"]) > 1e-2
    for k, r in rec["after"].items():
        exp = torch.tensor(r["full"]).reshape(r["shape"])
        err = (p[k] - exp).abs()
        if k.endswith("attn.qkv.bias"):
            D = exp.numel() // 3
            err[D:2 * D] = 0          # exactly-zero true gradient of the key bias (see golden_util)
        assert err.max() <= 2.5e-3, k
    # value / gradient of the term itself, incl. the zero-norm case (first step: param == global -> gradient 0)
    a = {"w": torch.tensor([3.0, 4.0]), "b": torch.tensor([1.0])}
    g0 = {"w": torch.tensor([0.0, 0.0]), "b": torch.tensor([1.0])}
    v, gr = O.prox_term(a, g0, 0.5, ["w", "b"])
    assert float(v) == pytest.approx(0.5 * 0.5 * 5.0) and torch.allclose(gr["w"], torch.tensor([0.15, 0.2])) and float(gr["b"].abs().max()) == 0


def test_c_abi_exports_every_declared_symbol():
    from fedcola_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fedcola_hip.h")).read()
    # the `#ifdef FC_PROBES` block declares what only the tools build exports (the round-5 experiments): checked against that library
    m = re.search(r"#ifdef FC_PROBES\n(.*?)#endif /\* FC_PROBES \*/", hdr, re.S)
    assert m, "the header's tools-build block is gone"
    product_hdr = hdr.replace(m.group(0), "")
    declared = sorted(set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", product_hdr)))
    probes_only = sorted(set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", m.group(1))))
    assert len(declared) >= 25 and set(probes_only) == set(_lib.PROBES_SIGNATURES), probes_only
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fedcola_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    for name in probes_only:
        assert not hasattr(lib, name), f"{name} is a tools-build experiment and must not be exported by the product library"
    assert lib.fc_abi_version() == 6
    probes = os.path.join(os.path.dirname(_lib.LIB_PATH), "libfedcola_hip_probes.so")
    if os.path.exists(probes):
        pl = ctypes.CDLL(probes)
        for name in declared + probes_only:
            assert hasattr(pl, name), f"{name} missing from the tools build"


def test_model_layout_and_errors_without_gpu():
    from fedcola_amd._lib import FedcolaHipError
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    for rec in G.load("init.json"):
        torch.manual_seed(rec["seed"])
        m = M(**rec["mk"])
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == rec["keys"]
        assert list(m.required_params().keys()) == rec["required"]
        for k, v in m.state_dict().items():
            G.check_summary(v, rec["sd"][k], 0.0, 0.0, f"init {rec['name']} {k}")      # bit-exact default init
    if not torch.cuda.is_available():
        with pytest.raises(FedcolaHipError):
            m([None, torch.zeros(2, 8, dtype=torch.long)])
    with pytest.raises(ValueError):
        M(modalities=["img", None], num_classes=[3, None], tasks=["cls", None], embed_dim=8, depth=1, num_heads=2, with_aux=True,
          aux_attn_only=True, aux_mlp_only=True)


def test_header_is_plain_c(tmp_path):
    """include/fedcola_hip.h is the drop-in boundary: it must compile as C99 (no C++ / torch types in the signatures)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "fedcola_hip.h"\nint main(void) { fc_model_cfg c; (void)c; return FC_ABI_VERSION == 6 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


# ---------------------------------------------------------------------------------------------- A1: sync_shared_weights / pretrain_vit
def _toy(modalities, **kw):
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    nc = [5 if modalities[0] == "img" else None, 3 if modalities[1] == "txt" else None] if None in modalities else [None, None]
    tasks = ["cls" if m is not None and None in modalities else ("rtv" if m is not None else None) for m in modalities]
    return M(modalities=modalities, num_classes=nc, tasks=tasks, embed_dim=8, depth=2, num_heads=2, vocab_size=30, max_text_len=8, **kw)


def test_scope_all_aliases_the_absent_tower_to_the_main_blocks():
    """mome.py:824-827: with share_scope == 'all' a uni-modal model's empty block slot IS the main tower's block list, so
    state_dict() lists blockses.{absent}.* keys that share storage with blockses.{main}.*; required_params() drops them again."""
    m = _toy(["img", None], share_scope="all")
    m.sync_shared_weights()
    sd = m.state_dict()
    alias = [k for k in sd if k.startswith("blockses.1.")]
    main = [k for k in sd if k.startswith("blockses.0.")]
    assert len(alias) == len(main) > 0
    for a, t in zip(alias, main):
        assert a == t.replace("blockses.0.", "blockses.1.", 1)
        assert sd[a].data_ptr() == sd[t].data_ptr()
    # reference order: the aliased slot follows every own key of the ModuleList walk only in position, not in content
    assert all("blockses.1." not in k for k in m.required_params())
    # named_parameters() de-duplicates shared tensors like nn.Module does
    assert all(not k.startswith("blockses.1.") for k, _ in m.named_parameters())
    # a checkpoint written from such a model loads back (alias keys accepted, same values)
    ck = {k: v.clone() + 1.0 for k, v in sd.items()}
    m.load_state_dict(ck, strict=True)
    assert torch.equal(m.state_dict()["blockses.0.0.attn.qkv.weight"], ck["blockses.1.0.attn.qkv.weight"])
    # scope != 'all': no alias keys
    m2 = _toy(["img", None], share_scope="modality")
    m2.sync_shared_weights()
    assert all(not k.startswith("blockses.1.") for k in m2.state_dict())


def test_colearn_param_blocks_is_the_reference_noop_and_attn_shares_modules():
    """mome.py:832-835 rebinds a loop variable only ('blocks' shares nothing); :836-840 ('attn') really aliases modules: the second
    tower's attention keys are views of the main tower's tensors, named_parameters() lists them once (init.json pins key order,
    parameter list and default init against the real reference)."""
    m = _toy(["img", "txt"], colearn_param="blocks")
    before = list(m.state_dict().keys())
    m.sync_shared_weights()
    assert list(m.state_dict().keys()) == before
    sd = m.state_dict()
    assert sd["blockses.0.0.attn.qkv.weight"].data_ptr() != sd["blockses.1.0.attn.qkv.weight"].data_ptr()
    m = _toy(["img", "txt"], colearn_param="attn")
    m.sync_shared_weights()
    sd = m.state_dict()
    for nm in ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias"):
        assert sd[f"blockses.0.0.{nm}"].data_ptr() == sd[f"blockses.1.0.{nm}"].data_ptr()
    assert sd["blockses.0.0.mlp.fc1.weight"].data_ptr() != sd["blockses.1.0.mlp.fc1.weight"].data_ptr()
    names = [k for k, _ in m.named_parameters()]
    assert "blockses.0.0.attn.qkv.weight" in names and "blockses.1.0.attn.qkv.weight" not in names
    keys = list(sd.keys())
    assert keys.index("blockses.1.0.norm1.bias") < keys.index("blockses.1.0.attn.qkv.weight") < keys.index("blockses.1.0.norm2.weight")
    # load_state_dict: the later key wins, like the reference's key-by-key copy into one shared tensor
    new = {k: torch.full_like(v, 2.0 if k.startswith("blockses.1.") else 1.0) for k, v in sd.items()}
    m.load_state_dict(new)
    assert float(m.state_dict()["blockses.0.0.attn.qkv.weight"].mean()) == 2.0


def test_pretrain_vit_key_mapping_on_a_synthetic_timm_state_dict():
    """mome.py:788-816 on a timm-shaped ViT state_dict (cls_token, pos_embed, patch_embed.proj.*, blocks.N.*, norm.*, head.*)."""
    m = _toy(["img", None])
    D, depth = 8, 2
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    timm_sd = {"cls_token": rnd(1, 1, D), "pos_embed": rnd(1, 197, D), "patch_embed.proj.weight": rnd(D, 3, 16, 16), "patch_embed.proj.bias": rnd(D),
               "norm.weight": rnd(D), "norm.bias": rnd(D), "head.weight": rnd(1000, D), "head.bias": rnd(1000)}
    for l in range(depth):
        for nm, shp in (("norm1.weight", (D,)), ("norm1.bias", (D,)), ("attn.qkv.weight", (3 * D, D)), ("attn.qkv.bias", (3 * D,)),
                        ("attn.proj.weight", (D, D)), ("attn.proj.bias", (D,)), ("norm2.weight", (D,)), ("norm2.bias", (D,)),
                        ("mlp.fc1.weight", (4 * D, D)), ("mlp.fc1.bias", (4 * D,)), ("mlp.fc2.weight", (D, 4 * D)), ("mlp.fc2.bias", (D,))):
            timm_sd[f"blocks.{l}.{nm}"] = rnd(*shp)
    seen = []
    own_head = m.state_dict()["heads.0.head.weight"].clone()
    m.pretrain_vit(["vit_small_patch16_224", None], loader=lambda name: (seen.append(name), dict(timm_sd))[1])
    assert seen == ["vit_small_patch16_224"]
    sd = m.state_dict()
    assert torch.equal(sd["embeddings.0.cls_token"], timm_sd["cls_token"]) and torch.equal(sd["embeddings.0.pos_embed"], timm_sd["pos_embed"])
    assert torch.equal(sd["embeddings.0.embed.proj.weight"], timm_sd["patch_embed.proj.weight"])
    assert torch.equal(sd["embeddings.0.embed.proj.bias"], timm_sd["patch_embed.proj.bias"])
    for l in range(depth):
        for nm in ("norm1.weight", "attn.qkv.weight", "attn.proj.bias", "mlp.fc1.weight", "mlp.fc2.bias", "norm2.bias"):
            assert torch.equal(sd[f"blockses.0.{l}.{nm}"], timm_sd[f"blocks.{l}.{nm}"]), (l, nm)
    assert torch.equal(sd["norm.weight"], timm_sd["norm.weight"])
    assert torch.equal(sd["heads.0.head.weight"], own_head)            # timm's 1000-way 'head.*' is not this model's 'heads.0.head.*'
    with pytest.raises(NotImplementedError):
        _toy(["img", None]).pretrain_vit(["vit_small_patch16_224", None])   # no checkpoint source offline: loud


def test_algorithm_plugin_fedavg_optimizer_is_fedavg():
    """src/algorithm/fedavg.py:7-55 (dormant plugin point): one accumulate per client with c = n_i / sum n, then step() = weighted mean."""
    from fedcola_amd.algorithm.fedavg import FedavgOptimizer
    from fedcola_amd.algorithm.fedprox import FedproxOptimizer
    g = torch.Generator().manual_seed(0)
    server = {"a.weight": torch.randn(4, 3, generator=g), "a.bias": torch.randn(4, generator=g), "bn.num_batches_tracked": torch.tensor(7.0)}
    clients = [{k: torch.randn(v.shape, generator=g) for k, v in server.items()} for _ in range(3)]
    sizes = [10, 30, 60]
    start = {k: v.clone() for k, v in server.items()}
    for cls in (FedavgOptimizer, FedproxOptimizer):
        for k in server:
            server[k].copy_(start[k])
        opt = cls(params=server, lr=1.0)
        for c, n in zip(clients, sizes):
            opt.accumulate({k: n / sum(sizes) for k in ("a.weight", "a.bias")}, iter(c.items()))
        opt.step()
        for k in ("a.weight", "a.bias"):
            exp = sum(c[k] * n for c, n in zip(clients, sizes)) / sum(sizes)
            assert float((server[k] - exp).abs().max()) < 1e-6
        assert float(server["bn.num_batches_tracked"]) == 7.0          # skipped by check_if
        opt.zero_grad()
        opt.step()                                                      # nothing pending: no change
        assert float((server["a.bias"] - sum(c["a.bias"] * n for c, n in zip(clients, sizes)) / sum(sizes)).abs().max()) < 1e-6


def test_fc_model_cfg_fields_agree_between_header_binding_and_integration_doc():
    """The C struct (include/fedcola_hip.h), the ctypes mirror (fedcola_amd/_lib.py) and the snippet a reference maintainer would copy
     must list the same fields in the same order: a binding made from a stale copy passes a short struct
    (the doc lacked colearn_attn)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "fedcola_hip.h")).read()
    body = re.search(r"typedef struct fc_model_cfg \{(.*?)\} fc_model_cfg;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        assert decl.startswith("int32_t "), decl
        c_fields += [f.strip() for f in decl[len("int32_t "):].split(",")]
    from fedcola_amd import _lib
    py_fields = [n for n, t in _lib.FcModelCfg._fields_]
    assert py_fields == c_fields
    import ctypes as C
    assert C.sizeof(_lib.FcModelCfg) == 4 * len(c_fields)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    snip = re.search(r"class FcModelCfg\(C\.Structure\):.*?_fields_ = \[\(n, C\.c_int32\) for n in \((.*?)\)\]", doc, re.S).group(1)
    doc_fields = re.findall(r'"(\w+)"', snip)
    assert doc_fields == c_fields
    # the same for fc_segment
    seg = re.search(r"typedef struct fc_segment \{(.*?)\} fc_segment;", hdr, re.S).group(1)
    seg = re.sub(r"/\*.*?\*/", "", seg, flags=re.S)
    seg_fields = [re.split(r"[\s\[]", d.strip().split()[-1])[0] for d in seg.split(";") if d.strip()]
    assert [n for n, t in _lib.FcSegment._fields_] == seg_fields


def test_refresh_from_equals_deepcopy_on_the_host():
    """ModalityAgnosticTransformer.refresh_from (what FedavgClient.download() uses to recycle a model object): same weights, freeze flags,
    mode and Python state as copy.deepcopy(src); refuses a model of another configuration without touching it."""
    import copy
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=4, depth=2, num_heads=2, vocab_size=30,
              max_text_len=8)
    torch.manual_seed(1)
    src, dst = M(**mk), M(**mk)
    with torch.no_grad():
        dst.flat.mul_(3.0).add_(1.0)
    k0 = list(src.segments)[3]
    src.set_trainable(k0, False)
    dst.eval()
    assert dst.refresh_from(src) is True
    ref = copy.deepcopy(src)
    assert dst.training == ref.training and dst.flat.data_ptr() != src.flat.data_ptr()
    for k, v in ref.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    assert [s["trainable"] for s in dst.segments.values()] == [s["trainable"] for s in ref.segments.values()]
    assert [n for n, p in dst.named_parameters()] == [n for n, p in ref.named_parameters()]
    other = M(**dict(mk, depth=3))
    before = other.flat.detach().clone()
    assert other.refresh_from(src) is False and torch.equal(other.flat, before)
    assert src.refresh_from(src) is False


def test_server_update_bookkeeping_matches_the_reference_rounds():
    """The host half of FedavgServer.update() against tests/golden/server_update.json (the REAL reference update(), three rounds): sampled
    ids incl. the warm-up filter (fedavgserver.py:307-308), the lr handed to each client (:509), requires_grad of every client parameter
    under freeze / unfreeze (:417-429, 511-516 -- the unfreeze includes aux_weight of an aux_trained=False model), LR decay (:851-852).
    Client training and the device blend are stubbed out here (no GPU); tests/test_gpu_fl.py runs the same rounds for real."""
    import random
    from collections import defaultdict
    import fl_util as F
    from fedcola_amd.client.fedavgclient import FedavgClient
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    from fedcola_amd.server.fedavgserver import FedavgServer
    from refstub import RefArgs
    rec = G.load("server_update.json")
    args = RefArgs(**F.ROUND_ARGS)
    srv = object.__new__(FedavgServer)
    srv.args, srv._round, srv.writer, srv.results, srv.curr_lr, srv.Cs = args, 0, None, defaultdict(dict), args.lr, dict(F.ROUND_CS)
    srv.global_models = {ds: M(with_aux=True, aux_trained=False, init=False, **F.round_model_kwargs(ds)) for ds in F.ROUND_DS}
    srv._init_param_scope(args.shared_param, args.share_scope)
    assert dict(srv.param_scope) == rec["scope"]
    trace = {}
    clients = []
    for cid, ds, n in F.ROUND_LAYOUT:
        task, mod = F.ROUND_DS[ds]
        cl = object.__new__(FedavgClient)
        cl._BaseClient__identifier, cl._BaseClient__model = cid, None
        cl.args, cl.dataset, cl.task, cl.modality, cl.training_set = args, ds, task, mod, list(range(n))

        def fake_update(cl=cl):
            trace[cl.id] = dict(lr=float(cl.args.lr), requires_grad={k: bool(p.requires_grad) for k, p in cl.model.named_parameters()})
            return {1: {"loss": 0.0, "metrics": {}}}
        cl.update = fake_update
        clients.append(cl)
    srv._clients = clients
    srv._aggregate = lambda *a, **k: None
    random.seed(F.ROUND_SEED)
    for r, exp in enumerate(rec["rounds"], start=1):
        srv.round = r
        trace.clear()
        assert srv.update() == exp["ids"]
        assert srv.curr_lr == pytest.approx(exp["curr_lr"], rel=1e-12)
        assert set(trace) == {int(k) for k in exp["clients"]}
        for cid, e in exp["clients"].items():
            assert trace[int(cid)]["lr"] == pytest.approx(e["lr"], rel=1e-12)
            assert trace[int(cid)]["requires_grad"] == e["requires_grad"], (r, cid)
        assert all(c.model is None for c in clients)


def test_server_local_evaluation_branch_and_small_surface_methods():
    """FedavgServer.evaluate with eval_type 'local' / 'both' (fedavgserver.py:858-869, 522-555): every client downloads the current global model,
    evaluates its hold-out set, and the results land under clients_evaluated_out; train_only skips it.  Plus the three small methods of the
    reference's surface: sync_shared_params (:160-168), _set_loaders (:170-171), _refine_optim_args (:432-440)."""
    from collections import defaultdict
    import fl_util as F
    from fedcola_amd.client.fedavgclient import FedavgClient
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    from fedcola_amd.server.fedavgserver import FedavgServer
    from refstub import RefArgs
    args = RefArgs(**dict(F.ROUND_ARGS, eval_type="both", train_only=False, R=3))
    srv = object.__new__(FedavgServer)
    srv.args, srv._round, srv.writer, srv.results = args, 3, None, defaultdict(dict)
    srv.global_models = {ds: M(with_aux=True, aux_trained=False, init=False, **F.round_model_kwargs(ds)) for ds in F.ROUND_DS}
    srv._init_param_scope(args.shared_param, args.share_scope)
    clients, downloaded = [], []
    for cid, ds, n in F.ROUND_LAYOUT:
        cl = object.__new__(FedavgClient)
        cl._BaseClient__identifier, cl._BaseClient__model = cid, None
        cl.args, cl.dataset, cl.task, cl.modality, cl.test_set = args, ds, *F.ROUND_DS[ds], list(range(n))
        cl.evaluate = lambda cl=cl: (downloaded.append((cl.id, cl.model is not None)), {"loss": 0.5 + cl.id, "metrics": {"acc1": 0.1 * cl.id}})[1]
        clients.append(cl)
    srv._clients = clients
    central = []
    srv._central_evaluate = lambda fedavg=False: central.append(fedavg)
    srv.evaluate([])
    assert [c for c, had in downloaded] == [c for c, _, _ in F.ROUND_LAYOUT] and all(had for _, had in downloaded) and central == [False]
    out = srv.results[3]["clients_evaluated_out"]
    assert set(out) == {str(c) for c, _, _ in F.ROUND_LAYOUT} and out["4"]["loss"] == 4.5
    assert all(c.model is None for c in clients)                                   # retain_model=False
    args.train_only = True
    downloaded.clear()
    srv.evaluate([])
    assert downloaded == []
    # sync_shared_params: attention keys (scope 'all' here) of every model become the LAST dataset's, the rest stays
    a = srv.global_models["CIFAR100"]
    last = srv.global_models[args.datasets[-1]]
    with torch.no_grad():
        last.flat.data.uniform_(-1, 1)
    before = {k: v.clone() for k, v in a.state_dict().items()}
    srv.sync_shared_params()
    lsd = last.state_dict()
    for k, v in a.required_params().items():
        if k in lsd and srv.param_scope[k] != "dataset":
            assert torch.equal(v, lsd[k]), k
        else:
            assert torch.equal(v, before[k]), k
    srv._set_loaders((None, {"x": 1}))
    assert srv.server_dataset == {"x": 1}
    args.optimizer = "SGD"
    assert srv._refine_optim_args(args) == {"lr": args.lr, "momentum": args.momentum, "weight_decay": args.weight_decay, "nesterov": args.nesterov}
