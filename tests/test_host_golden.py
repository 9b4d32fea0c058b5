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
    declared = sorted(set(re.findall(r"\b(fc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fedcola_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.fc_abi_version() == 1


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
    src.write_text('#include "fedcola_hip.h"\nint main(void) { fc_model_cfg c; (void)c; return FC_ABI_VERSION == 1 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
