"""Model-level parity on the GPU: the HIP client step (through the C ABI) against the oracle and the golden vectors."""
import os
import pytest
import torch

import golden_util as G
import product_util as PU
from oracle import mome_oracle as O
from test_oracle_golden import cfg_from_mk

pytestmark = pytest.mark.gpu

CASES = ["toy", "small", "imgcls_aux", "txtcls_aux", "colearn_attn"]


def grad_tol(k, go, grads_o, weights, rtol, atol):
    """Tolerance for one gradient tensor.  cross_modal_scale's gradient is the scalar <dW_eff, A>: a sum with heavy
    cancellation, so its error bound scales with sum|dW_eff*A|, not with the (small) result."""
    if k.endswith("cross_modal_scale"):
        wk = k.replace("cross_modal_scale", "weight")
        ak = k.replace("cross_modal_scale", "aux_weight")
        return rtol * float((grads_o[wk] * weights[ak]).abs().sum()) + atol
    return rtol * max(float(go.abs().max()), 1e-6) + atol


def oracle_step(case, rec, dp_masks=None):
    cfg = cfg_from_mk(rec["mk"])
    p = O.resolve_colearn(G.case_weights(case), cfg)
    img, ids, y = G.case_inputs(rec)
    batch = {"img+txt": ("img+txt", img, ids), "img": ("img", img, y), "txt": ("txt", ids, y)}[rec["kind"]]
    state = dict(step=0, m={}, v={})
    loss, outs, grads = O.client_step(p, cfg, batch, state, lr=rec["lr"], dp_masks=dp_masks)
    return cfg, p, float(loss), outs, grads


@pytest.mark.parametrize("case", CASES)
def test_client_step_fp32_vs_oracle_and_golden(case):
    rec = G.load(f"model_{case}.json")
    cfg, p_after, loss_o, outs_o, grads_o = oracle_step(case, rec)
    img, ids, y = G.case_inputs(rec)
    model = PU.build_product(rec["mk"], "fp32", G.case_weights(case))
    # forward outputs (autograd-free path)
    model.train()
    with torch.no_grad():
        outs = model([img.cuda() if rec["kind"] != "txt" else None, ids.cuda() if rec["kind"] != "img" else None],
                     feat_out=rec["kind"] == "img+txt")
    for o, oo, r in zip(outs, outs_o, rec["outs"]):
        if oo is None:
            assert o is None
            continue
        assert (o.cpu() - oo).abs().max() <= 1e-4 * max(1.0, float(oo.abs().max())), "outs vs oracle"
        G.compare(o, r, 1e-4, 1e-6, f"{case} outs vs golden")
    loss, grads, st = PU.product_step(model, rec["kind"], img, ids, y, rec["lr"])
    assert abs(loss - loss_o) <= 1e-4 * max(1.0, abs(loss_o))
    assert abs(loss - rec["loss"]) <= 1e-4 * max(1.0, abs(rec["loss"]))
    w0 = G.case_weights(case)
    for k, go in grads_o.items():
        err = float((grads[k] - go).abs().max())
        assert err <= grad_tol(k, go, grads_o, w0, 1e-4, 1e-7), f"grad {k}: err {err} scale {float(go.abs().max())}"
        if rec["grads"][k] is not None and not k.endswith("cross_modal_scale"):
            G.compare(grads[k], rec["grads"][k], 3e-4, 1e-7, f"{case} grad {k} vs golden")
    sd = model.state_dict()
    for k, r in rec["after"].items():
        G.compare_after_adamw(sd[k], r, rec["grads"][k], rec["lr"], f"{case} after {k}")


@pytest.mark.parametrize("case", CASES)
def test_client_step_bf16_close_to_oracle(case):
    """bf16 storage / MFMA mode: stated tolerance 6e-2 of each gradient tensor's max (bf16 has 8 mantissa bits and
    errors compound over depth); loss within 3e-2."""
    rec = G.load(f"model_{case}.json")
    cfg, p_after, loss_o, outs_o, grads_o = oracle_step(case, rec)
    img, ids, y = G.case_inputs(rec)
    model = PU.build_product(rec["mk"], "bf16", G.case_weights(case))
    model.train()
    loss, grads, st = PU.product_step(model, rec["kind"], img, ids, y, rec["lr"])
    assert abs(loss - loss_o) <= 3e-2 * max(1.0, abs(loss_o))
    w0 = G.case_weights(case)
    for k, go in grads_o.items():
        err = float((grads[k] - go).abs().max())
        assert err <= grad_tol(k, go, grads_o, w0, 6e-2, 1e-6), f"bf16 grad {k}: err {err} scale {float(go.abs().max())}"


def test_autograd_path_and_droppath_masks():
    """loss.backward() through the autograd.Function, with explicit DropPath multipliers, vs the oracle."""
    case = "small"
    rec = G.load(f"model_{case}.json")
    img, ids, y = G.case_inputs(rec)
    B = rec["B"]
    depth = rec["mk"]["depth"]
    dp = torch.ones(2, depth, 2, B)
    dp[0, 1, 0] = torch.tensor([2.0, 0.0, 2.0, 0.0]); dp[0, 1, 1] = torch.tensor([0.0, 2.0, 2.0, 0.0])
    dp[1, 0, 1] = torch.tensor([1.25, 1.25, 0.0, 1.25]); dp[1, 1, 0] = torch.tensor([0.0, 0.0, 2.0, 2.0])
    masks = {(t, l, br): dp[t, l, br] for t in range(2) for l in range(depth) for br in range(2)}
    cfg, p_after, loss_o, outs_o, grads_o = oracle_step(case, rec, dp_masks=masks)
    model = PU.build_product(rec["mk"], "fp32", G.case_weights(case))
    model.train()
    outs = model([img.cuda(), ids.cuda()], feat_out=True, droppath=dp.cuda().contiguous())
    tau = O.contrastive_tau()
    Lg = tau * outs[0] @ outs[1].t()
    lab = torch.arange(B, device="cuda")
    loss = 0.5 * (torch.nn.functional.cross_entropy(Lg, lab) + torch.nn.functional.cross_entropy(Lg.t(), lab))
    loss.backward()
    assert abs(float(loss) - loss_o) <= 1e-4
    named = dict(model.named_parameters())
    for k, go in grads_o.items():
        scale = max(float(go.abs().max()), 1e-6)
        err = float((named[k].grad.cpu() - go).abs().max())
        assert err <= 1e-4 * scale + 1e-7, f"grad {k}: err {err} scale {scale}"


def test_errors_are_loud():
    from fedcola_amd._lib import FedcolaHipError
    rec = G.load("model_imgcls_aux.json")
    model = PU.build_product(rec["mk"], "fp32", G.case_weights("imgcls_aux"))
    with pytest.raises(AssertionError):
        model([None, torch.zeros(2, 8, dtype=torch.long).cuda()])        # None modality should have None input
    with pytest.raises(AssertionError):
        model([torch.zeros(2, 3, 32, 32).cuda(), None])                  # image size mismatch
    cpu_model = PU.build_product(rec["mk"], "fp32", G.case_weights("imgcls_aux")).cpu()
    with pytest.raises(FedcolaHipError):
        cpu_model([torch.zeros(2, 3, 224, 224), None])                   # no CPU fallback


@pytest.mark.parametrize("B", [17, 25, 37])
@pytest.mark.parametrize("kind", ["img+txt", "img"])
def test_microbatch_chains_do_not_change_the_result(tmp_path, kind, B):
    """The image tower runs as micro-batch chains by default (FC_MICROBATCH, read once per process), with or without a text tower
    beside it: an odd batch (B = 17 -> 10 + 7 in both directions; B = 25 / 37 -> three forward chains cut at B/3, 2B/3 and, beside a text
    tower, three backward chains with the text tower on the weight-gradient stream -- two backward chains cut at 57 % for the image-only
    client) gives the same gradients as the single-chain run up to the order of the LayerNorm partial sums.
    'img' = an image classifier with trained re-param linears (its head's weight gradients are taken for the full batch before the
    chains fork)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    if not os.path.exists(os.path.join(root, "fedcola_amd", "libfedcola_hip_probes.so")):
        pytest.skip("tools build (python -m fedcola_amd.build --probes) not present: the product library reads no tuning knobs")
    for mb in ("1", "2"):
        f = str(tmp_path / f"mb{mb}.pt")
        env = dict(os.environ, FC_MICROBATCH=mb, FC_PROBES_LIB="1")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "mb_check.py"), str(B), f, kind], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mb] = torch.load(f)
    worst = {"matrix": ("", 0.0), "vector": ("", 0.0), "atomic": ("", 0.0)}
    for k in out["1"]:
        scale = max(float(out["1"][k].abs().max()), 1e-9)
        err = float((out["1"][k] - out["2"][k]).abs().max()) / scale
        # Every row-wise tensor of the step is the same bits in both schedules; what differs is the grouping of rows in the column sums.
        # matrix: weight gradients, full-batch weight-gradient problems in either schedule.  vector: the linears' biases (column sums
        # inside those problems) and the blocks' LayerNorm weight / bias (fp32 block partial rows of 32 rows each, added in fp64 by
        # k_ln_reduce).  atomic: embedding tables, position / type rows, the shared final norm, a classification head and the re-param
        # scalars -- summed with fp32 atomics whose order is not fixed: equal up to the order of the sum.
        cls = "atomic" if ("embeddings" in k or k.startswith(("norm.", "heads.")) or k.endswith("cross_modal_scale")) else ("matrix" if out["1"][k].dim() > 1 else "vector")
        worst[cls] = max(worst[cls], (k, err), key=lambda t: t[1])
    print(f"schedule invariance {kind} B={B}: {worst}")
    # measured (round 4, six cases, two builds): matrix 0 (the same bits), vector <= 2.5e-7, atomic <= 9.6e-6 (a cross_modal_scale)
    assert worst["matrix"][1] <= 1e-7, worst["matrix"]
    assert worst["vector"][1] <= 2e-5, worst["vector"]
    assert worst["atomic"][1] <= 1e-4, worst["atomic"]


@pytest.mark.parametrize("kind,width,dw_wide,B", [("img+txt", 128, "2", 0), ("img", 128, "2", 0), ("img+txt", 384, "2", 0), ("img+txt", 384, "1", 0),
                                                  ("img+txt", 384, "2", 32)])
def test_fused_optimizer_is_bit_identical_to_the_separate_one(tmp_path, kind, width, dw_wide, B):
    """fc_client_step takes the AdamW step of the linears inside the weight-gradient GEMM's epilogue (FC_FUSED_OPT, tools build, read
    once per process, default on).  One step with weight decay from non-trivial moments: the linears' parameters, both moments, bf16
    compute weights and gradients are the same BITS as with the separate optimizer pass.  Width 128: the 128x128-tile grouped kernel;
    width 384 (depth 2, B = 16, two chains): the wide 128x384-tile kernels of the ViT-S / ViT-B steps, in both of their forms
    (FC_DW_WIDE = 2: 8 consumer + 2 loader waves, the default; 1: 8 waves).  B = 32: three image chains in both directions, text tower on the
    weight-gradient stream, the last chunk on a chain's stream -- the separate optimizer's two phases around that chunk included."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fedcola_amd", "libfedcola_hip_probes.so")):
        pytest.skip("tools build (python -m fedcola_amd.build --probes) not present: the product library reads no tuning knobs")
    out = {}
    for fused in ("0", "1"):
        f = str(tmp_path / f"f{fused}.pt")
        env = dict(os.environ, FC_FUSED_OPT=fused, FC_PROBES_LIB="1", FC_DW_WIDE=dw_wide)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "opt_check.py"), kind, f, str(width)] + ([str(B)] if B else []), env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[fused] = torch.load(f)
    assert abs(out["0"]["loss"] - out["1"]["loss"]) <= 1e-5          # the loss sum is atomic: equal up to its order
    n_lin = 0
    for name, (o, c) in out["0"]["segs"].items():
        linear = ".attn." in name or ".mlp." in name
        for k in ("g", "p", "m", "v"):
            a, b = out["0"][k][o:o + c], out["1"][k][o:o + c]
            if linear:      # stepped inside the weight-gradient epilogue
                assert torch.equal(a, b), f"{k} of {name}: {int((a != b).sum())} of {c} elements differ"
            else:           # embedding / LayerNorm gradients are summed with atomics: equal up to the order of the sum
                assert float((a - b).abs().max()) <= 1e-4 * max(float(a.abs().max()), 1e-12), f"{k} of {name}"
        if linear:
            n_lin += c
            wa, wb = out["0"]["wc"][o:o + c], out["1"]["wc"][o:o + c]
            assert torch.equal(wa, wb), f"bf16 compute weights of {name}"
    assert n_lin > 0.8 * out["0"]["p"].numel()


def test_device_table_cache_survives_starting_over(tmp_path):
    """The device tables of the grouped launches are cached per process by content and the cache starts over when it is full
    (fc_model.hip::cached_table).  With a limit of 3 entries (FC_TABLE_CACHE_MAX, tools build) and batch sizes that keep changing the
    cache starts over several times per run, also between a step's weight-gradient chunks and its optimizer remainder: no handle may
    keep a cache-owned pointer across that (the optimizer's chunk table did).  Same parameters as with the default limit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fedcola_amd", "libfedcola_hip_probes.so")):
        pytest.skip("tools build (python -m fedcola_amd.build --probes) not present: the product library reads no tuning knobs")
    out = {}
    for lim in ("3", "1024"):
        f = str(tmp_path / f"lim{lim}.pt")
        env = dict(os.environ, FC_TABLE_CACHE_MAX=lim, FC_PROBES_LIB="1")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "table_cache_check.py"), f], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[lim] = torch.load(f)
    a, b = out["3"]["p"], out["1024"]["p"]
    assert bool(torch.isfinite(a).all())
    # eight AdamW steps of 1e-3: a parameter moves by <= 8e-3; run-to-run noise of the atomically summed embedding gradients can flip
    # the sign of a step for near-zero gradients, a stale table corrupts whole chunks
    assert float((a - b).abs().max()) <= 4e-3 and float((a - b).abs().mean()) <= 2e-5, (float((a - b).abs().max()), float((a - b).abs().mean()))
    for x, y in zip(out["3"]["losses"], out["1024"]["losses"]):
        assert abs(x - y) <= 2e-2 * max(1.0, abs(y))


@pytest.mark.parametrize("B", [8, 32])
def test_colearn_attn_d384_bf16_shared_attention_gradients(B):
    """colearn_param == 'attn' at the ViT-S width in the timed mode: the text tower's weight gradients of the shared qkv / proj go
    through the 128x384 grouped kernel into the side buffer and are added to the image tower's; against the emulating oracle.
    B = 8: one image chain; B = 32: three image chains in both directions with the text tower on the weight-gradient stream."""
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    from synth import det_state_dict
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=384, depth=2, num_heads=6,
              vocab_size=97, max_text_len=16, colearn_param="attn")
    cfg = cfg_from_mk(mk)
    torch.manual_seed(3)
    ref = M(**mk)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    seq = 16
    g = torch.Generator().manual_seed(21)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
    ids = torch.randint(1, 97, (B, seq), generator=g)
    p = O.resolve_colearn({k: v.clone() for k, v in sd.items()}, cfg)
    with O.emulate_bf16():
        outs_o, cache = O.forward(p, cfg, [img, ids], feat_out=True)
        loss_o, da, db = O.contrastive_loss(outs_o[0], outs_o[1])
        grads_o = O.backward(p, cfg, cache, [da, db])
    model = PU.build_product(mk, "bf16", sd)
    model.train()
    loss, grads, _ = PU.product_step(model, "img+txt", img, ids, None, 1e-4)
    assert abs(loss - float(loss_o)) <= 3e-2 * max(1.0, abs(float(loss_o)))
    for k in ("blockses.0.0.attn.qkv.weight", "blockses.0.1.attn.proj.weight", "blockses.0.1.attn.qkv.bias", "blockses.1.0.mlp.fc1.weight"):
        go = grads_o[k]
        rel = float((grads[k] - go).norm() / go.norm())
        assert rel <= 8e-2, (k, rel)
    assert "blockses.1.0.attn.qkv.weight" not in grads


@pytest.mark.probes
@pytest.mark.parametrize("B", [16, 32])
def test_step_graph_replay_is_bit_identical_to_the_eager_step(B):
    """fc_client_step replays a captured HIP graph from the third step with the same buffers on (fc_model.hip: whole-step graph).  From one
    saved state, the captured step, a pure replay and an eager step (another loss buffer = another key = never captured) must produce the
    same bits for everything that is not summed with atomics, with a learning rate and a step count that differ from the captured ones
    (the AdamW constants of a replay come from device memory, not from the baked kernel arguments)."""
    import product_util as PU
    from synth import det_state_dict
    from fedcola_amd import _lib
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=384, depth=2, num_heads=6, vocab_size=64, max_text_len=16)
    torch.manual_seed(0)
    sd = det_state_dict({k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}, base_seed=5)
    g = torch.Generator().manual_seed(3)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1).cuda()
    ids = torch.randint(1, 64, (B, 16), generator=g).cuda()
    model = PU.build_product(mk, "bf16", sd); model.train()
    model.set_option("step_graph", 1)
    n = model.flat.numel()
    st = dict(g=torch.zeros(n).cuda(), m=torch.zeros(n).cuda(), v=torch.zeros(n).cuda())
    loss_a, loss_b = torch.zeros(2).cuda(), torch.zeros(2).cuda()
    model.prepare_weights(force=True)
    ws = model.workspace(B, 16)
    P, L = _lib.ptr, _lib.lib()

    own = torch.cuda.Stream()      # a stream capture cannot begin on the legacy default stream (the capture then declines: what ADVICE r05 suspected)
    torch.cuda.synchronize()

    def step(k, lr, lossbuf, b1=0.9, b2=0.999):
        with torch.cuda.stream(own):
            _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(st["g"]), P(st["m"]), P(st["v"]), P(model._wc_or_flat()), P(img), P(ids), None, B, 16,
                                        None, lr, b1, b2, 1e-8, 0.01, k, P(lossbuf), P(ws), ws.numel(), _lib.stream_ptr()))
        torch.cuda.synchronize()

    def snapshot():
        return dict(p=model.flat.detach().clone(), wc=model._wc_or_flat().detach().clone(), **{k: v.clone() for k, v in st.items()})

    def restore(s):
        model.flat.data.copy_(s["p"]); model._wc_or_flat().copy_(s["wc"])
        for k in st: st[k].copy_(s[k])

    step(1, 1e-3, loss_a); step(2, 1e-3, loss_a)          # two eager steps with these buffers ...
    s0 = snapshot()
    outs = {}
    for name, k, lr, buf in (("captured", 3, 1e-3, loss_a), ("replay", 7, 3e-3, loss_a), ("eager", 7, 3e-3, loss_b), ("replay2", 7, 3e-3, loss_a)):
        restore(s0)
        buf.zero_()
        step(k, lr, buf)                                    # ... the third one is captured, later ones with loss_a are replays
        outs[name] = dict(snapshot(), loss=float(buf[1]))
    segs = model.segments
    for a, b in (("replay", "eager"), ("replay2", "replay")):
        assert abs(outs[a]["loss"] - outs[b]["loss"]) <= 1e-5 * max(1.0, abs(outs[b]["loss"]))
        n_lin = 0
        for name, sg in segs.items():
            o, c = int(sg["offset"]), int(sg["numel"])
            linear = ".attn." in name or ".mlp." in name
            for key in ("g", "p", "m", "v"):
                x, y = outs[a][key][o:o + c], outs[b][key][o:o + c]
                if linear:
                    assert torch.equal(x, y), f"{a} vs {b}: {key} of {name}: {int((x != y).sum())} of {c} elements differ"
                else:      # embedding / LayerNorm gradients are summed with atomics: equal up to the order of the sum
                    assert float((x - y).abs().max()) <= 1e-4 * max(float(x.abs().max()), 1e-12), f"{a} vs {b}: {key} of {name}"
            n_lin += c if linear else 0
        assert n_lin > 0.8 * n
    # the replayed step really used lr = 3e-3 at step 7, not the captured 1e-3 at step 3
    assert not torch.equal(outs["captured"]["p"], outs["replay"]["p"])
    # ... and the steps that were meant to be graph launches WERE: the capture (1) and the two replays; a capture that silently declines would
    # pass every comparison above on eager steps
    assert int(L.fc_dbg_step_graph_hits(model._handle.h)) == 3, L.fc_last_error()
    # other betas are another graph: the captured kernels bake them in (the key holds them since round 6)
    restore(s0)
    step(7, 3e-3, loss_a, b1=0.8, b2=0.99)
    assert int(L.fc_dbg_step_graph_hits(model._handle.h)) == 3 and not torch.equal(model.flat.detach(), outs["replay"]["p"])


@pytest.mark.probes
@pytest.mark.parametrize("option,value", [("mlp_fused", 1), ("gemm_form", 64), ("gemm_form", 3), ("gemm_form", 4)])
def test_optional_kernel_forms_give_the_same_step(option, value):
    """The forms fc_model_set_option switches on (fused MLP, 64-row GEMM tiles with or without the deep staging ring) are other schedules of
    the same arithmetic: every product sums its k in the same order, so one client step from the same state gives the same bits in every
    linear's gradient / parameters / moments (embedding and LayerNorm gradients are summed with atomics: equal up to the order).
    ViT-S width, depth 2, B = 22 (three image chains of 7-8 samples beside the text tower: the under-filled launches the forms are for)."""
    import product_util as PU
    from synth import det_state_dict
    from fedcola_amd import _lib
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    mk = dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], embed_dim=384, depth=2, num_heads=6, vocab_size=64, max_text_len=16)
    torch.manual_seed(0)
    sd = det_state_dict({k: tuple(v.shape) for k, v in M(**mk).state_dict().items()}, base_seed=5)
    B = 22
    g = torch.Generator().manual_seed(3)
    img = (torch.randn(B, 3, 224, 224, generator=g) * 0.5).clamp_(-1, 1)
    ids = torch.randint(1, 64, (B, 16), generator=g)
    outs = {}
    try:
        for tag in ("default", "option"):
            model = PU.build_product(mk, "bf16", sd); model.train()
            if tag == "option":
                model.set_option(option, value)
            n = model.flat.numel()
            g2 = torch.Generator().manual_seed(11)
            st = dict(grads=torch.zeros(n).cuda(), m=(torch.randn(n, generator=g2) * 1e-3).cuda(), v=(torch.rand(n, generator=g2) * 1e-5).cuda(), loss=torch.zeros(2).cuda())
            loss, _, st = PU.product_step(model, "img+txt", img, ids, None, 1e-3, wd=0.01, step=3, state=st)
            outs[tag] = dict(p=model.flat.detach().cpu(), m=st["m"].cpu(), v=st["v"].cpu(), g=st["grads"].cpu(), loss=loss, segs=model.segments)
    finally:
        if option == "gemm_form":
            _lib.check(_lib.lib().fc_model_set_option(model._handle.h, _lib.FC_OPT_GEMM_FORM, 0))      # process-wide: back to the default
    a, b = outs["default"], outs["option"]
    assert abs(a["loss"] - b["loss"]) <= 1e-5 * max(1.0, abs(a["loss"]))
    n_lin = 0
    for name, sg in a["segs"].items():
        o, c = int(sg["offset"]), int(sg["numel"])
        linear = ".attn." in name or ".mlp." in name
        for key in ("g", "p", "m", "v"):
            x, y = a[key][o:o + c], b[key][o:o + c]
            if linear and option == "gemm_form":
                assert torch.equal(x, y), f"{key} of {name}: {int((x != y).sum())} of {c} elements differ"
            else:      # atomically summed gradients; the fused MLP's residual add may contract differently (a rare one-ulp bf16 flip upstream)
                assert float((x - y).abs().max()) <= 2e-3 * max(float(x.abs().max()), 1e-12), f"{key} of {name}"
        n_lin += c if linear else 0
    assert n_lin > 0.8 * a["p"].numel()


def test_the_product_library_has_no_run_time_options():
    """the experiments that do not pay (fused MLP, 64-row / ring / split GEMM forms, whole-step graph) left the product
    library; set_option says where they went instead of silently doing nothing."""
    import product_util as PU
    from fedcola_amd import _lib
    if _lib.is_probes_build():
        pytest.skip("running on the tools build")
    rec = G.load("model_toy.json")
    model = PU.build_product(rec["mk"], "fp32", G.case_weights("toy"))
    for name in ("mlp_fused", "step_graph", "gemm_form"):
        with pytest.raises(_lib.FedcolaHipError, match="tools build"):
            model.set_option(name, 1)
    assert not hasattr(_lib.lib(), "fc_k_mlp_fused") and not hasattr(_lib.lib(), "fc_model_set_option")


def test_tools_build_experiments():
    """The tests marked `probes` (fused MLP kernel and in-model form, GEMM forms, whole-step graph replay incl. its hit counter) against
    libfedcola_hip_probes.so, in a child process (FC_PROBES_LIB is read when fedcola_amd._lib is imported)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fedcola_amd", "libfedcola_hip_probes.so")):
        pytest.skip("tools build absent (python -m fedcola_amd.build --probes)")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu and probes", os.path.join(root, "tests", "test_gpu_model.py"),
                        os.path.join(root, "tests", "test_gpu_kernels.py")], env=dict(os.environ, FC_PROBES_LIB="1"), capture_output=True, text=True,
                       timeout=1500, cwd=root)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail, tail
