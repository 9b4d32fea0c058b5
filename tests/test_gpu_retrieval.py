"""N1 on the GPU: the fp64 similarity GEMM and the rank kernel (through the C ABI) against the oracle and against the
golden vectors of the reference's COCOEvaluator; size-independent properties at Flickr30k / COCO-5k sizes."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import retrieval_oracle as ro
from retrieval_util import RETRIEVAL_CASES, FakeDataset, FakeLoader, PassThroughModel, retrieval_set, stream, unit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "retrieval.json")))


def ranks_hip(q, g, ql, gl, batch_size=1024):
    from fedcola_amd.metrics.eval_coco import best_ranks_device
    return best_ranks_device(q, g, ql, gl, "cuda", batch_size)


@pytest.mark.parametrize("nq,ng,d", [(1, 1, 1), (3, 5, 7), (64, 64, 16), (65, 130, 33), (200, 77, 384), (130, 257, 50)])
def test_sim_f64_matches_numpy(nq, ng, d):
    from fedcola_amd import _lib
    gen = torch.Generator().manual_seed(nq * 1000 + ng * 10 + d)
    q = torch.randn(nq, d, generator=gen, dtype=torch.float64)
    g = torch.randn(ng, d, generator=gen, dtype=torch.float64)
    qd, gd = q.cuda(), g.cuda()
    S = torch.full((nq, ng), float("nan"), dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().fc_k_sim_f64(_lib.ptr(qd), _lib.ptr(gd), _lib.ptr(S), nq, ng, d, _lib.stream_ptr()))
    torch.cuda.synchronize()
    ref = q.numpy() @ g.numpy().T
    assert np.abs(S.cpu().numpy() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max()) * d ** 0.5     # fp64: tolerance stated


@pytest.mark.parametrize("n_img,caps,d,seed,batch", [(7, 1, 5, 1, 1024), (33, 5, 16, 2, 1024), (100, 5, 64, 3, 17), (64, 3, 384, 4, 64)])
def test_ranks_bit_exact_vs_oracle(n_img, caps, d, seed, batch):
    img, cap, iids, _ = retrieval_set(n_img, caps, d, seed)
    cls_img = iids.numpy().astype(np.float64)
    cls_cap = np.repeat(cls_img, caps)
    for q, g, ql, gl in ((img, cap, cls_img, cls_cap), (cap, img, cls_cap, cls_img)):
        got = ranks_hip(q, g, ql, gl, batch)
        exp = ro.best_ranks(q.double().numpy(), g.double().numpy(), ql, gl)
        assert np.array_equal(got, exp)


def test_ties_resolve_by_gallery_index():
    # duplicated gallery rows (exact ties, also between a positive and a negative) and duplicated queries
    base = unit(torch.randn(6, 8, generator=torch.Generator().manual_seed(0), dtype=torch.float64))
    g = torch.cat([base, base, base[:2]])                       # 14 rows, many exact duplicates
    gl = np.array([0, 1, 2, 3, 4, 5, 5, 4, 3, 2, 1, 0, 9, 1], dtype=np.float64)
    q = torch.cat([base, base[:3]])
    ql = np.array([0, 1, 2, 3, 4, 5, 1, 9, 2], dtype=np.float64)
    got = ranks_hip(q, g, ql, gl)
    exp = ro.best_ranks(q.numpy(), g.numpy(), ql, gl)
    assert np.array_equal(got, exp)
    assert got[0] == 0 and got[7] > 0


def test_query_without_positive_raises_like_reference():
    q = torch.eye(3, dtype=torch.float64)
    with pytest.raises(ValueError):
        ranks_hip(q, q, np.array([1.0, 2.0, 7.0]), np.array([1.0, 2.0, 3.0]))
    from fedcola_amd._lib import FedcolaHipError
    from fedcola_amd.metrics.eval_coco import best_ranks_device
    with pytest.raises(FedcolaHipError):
        best_ranks_device(q, q, np.arange(3.0), np.arange(3.0), "cpu")     # no CPU fallback


@pytest.mark.parametrize("name", list(RETRIEVAL_CASES))
def test_evaluator_matches_reference_golden(name):
    """COCOEvaluator.evaluate end to end (features on cuda, HIP ranking) == the reference evaluator's scores and ranks."""
    from fedcola_amd.metrics.eval_coco import COCOEvaluator
    c, gold = RETRIEVAL_CASES[name], GOLD[name]
    img, cap, iids, aids = retrieval_set(c["n_images"], c["caps"], c["D"], c["seed"])
    batches = stream(img, cap, iids, aids, c["caps"], c["batch"])
    ev = COCOEvaluator("matmul", n_crossfolds=c["folds"], extract_device="cuda", eval_device="cuda")
    ev.set_model(PassThroughModel(c["D"]))
    loader = FakeLoader(batches, FakeDataset(c["n_images"], cap.shape[0]))
    ex = ev.extract_features(loader)
    assert [int(v) for v in ex["image_ids"]] == gold["image_ids"]
    assert [int(v) for v in ex["caption_ids"]] == gold["caption_ids"]
    r_i2t = ranks_hip(ex["image_features"], ex["caption_features"], ex["image_classes"], ex["caption_classes"])
    r_t2i = ranks_hip(ex["caption_features"], ex["image_features"], ex["caption_classes"], ex["image_classes"])
    assert [int(r) for r in r_i2t] == gold["ranks_i2t"]
    assert [int(r) for r in r_t2i] == gold["ranks_t2i"]
    scores = ev.evaluate(loader, n_images_per_crossfold=c["ipf"], n_captions_per_crossfold=c["cpf"], eval_batch_size=64)

    def check(got, exp):
        for k, v in exp.items():
            if isinstance(v, dict):
                check(got[k], v)
            else:
                assert float(got[k]) == pytest.approx(v, rel=1e-12, abs=1e-12), k
    check(scores, gold["scores"])


@pytest.mark.parametrize("n_img,caps,d", [(1000, 5, 384), (5000, 5, 384)])       # Flickr30k test split / COCO 5k
def test_full_size_properties(n_img, caps, d):
    gen = torch.Generator().manual_seed(n_img)
    img = unit(torch.randn(n_img, d, generator=gen))
    cap = unit(img.repeat_interleave(caps, 0) + 0.35 * torch.randn(n_img * caps, d, generator=gen))
    li = np.arange(n_img, dtype=np.float64)
    lc = np.repeat(li, caps)
    r_t2i = ranks_hip(cap, img, lc, li)
    r_i2t = ranks_hip(img, cap, li, lc)
    # (a) spot-check against the oracle on a slice of queries (full gallery)
    sl = np.arange(0, n_img * caps, max(1, n_img * caps // 97))
    assert np.array_equal(r_t2i[sl], ro.best_ranks(cap[sl].double().numpy(), img.double().numpy(), lc[sl], li))
    si = np.arange(0, n_img, max(1, n_img // 61))
    assert np.array_equal(r_i2t[si], ro.best_ranks(img[si].double().numpy(), cap.double().numpy(), li[si], lc))
    # (b) a query that IS its positive ranks first; every rank is inside the gallery
    assert np.all(ranks_hip(img, img, li, li) == 0)
    assert r_t2i.min() >= 0 and r_t2i.max() < n_img and r_i2t.max() < n_img * caps
    # (c) permuting the gallery does not change the ranks (no exact ties in random data); query batching does not either
    perm = torch.randperm(n_img, generator=gen)
    assert np.array_equal(ranks_hip(cap, img[perm], lc, li[perm.numpy()], batch_size=4096), r_t2i)
    # (d) the best rank over all positives is <= the rank of one given positive (same gallery, the others relabelled)
    lc1 = lc.copy()
    lc1[np.arange(len(lc)) % caps != 0] = -1.0
    assert np.all(r_i2t <= ranks_hip(img, cap, li, lc1))


def test_evaluator_on_the_hip_model():
    """extract_features through the product model (HIP forward, feat_out=True) + HIP ranking == oracle ranking of the same features."""
    from fedcola_amd.metrics.eval_coco import COCOEvaluator
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    from synth import det_ids, det_tensor
    torch.manual_seed(3)
    m = M(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], img_size=32, patch_size=16, embed_dim=64, depth=2,
          num_heads=2, mlp_ratio=2, vocab_size=97, max_text_len=8, precision="fp32").cuda()
    n_img, caps = 12, 5
    images = det_tensor([n_img, 3, 32, 32], 11, 1.0)
    tokens = det_ids([n_img * caps, 8], 5, 97)
    tokens[:, 0] = 1 + torch.arange(n_img * caps)      # det_ids repeats with period 12: make every caption distinct, so that no two gallery
                                                       # entries have EQUAL similarity to a query (the tie order is pinned by its own tests above;
                                                       # here a tie would compare numpy's dgemm rounding of duplicate columns with the kernel's)
    iids = 100 + 3 * torch.arange(n_img)
    aids = 7000 + torch.arange(n_img * caps)
    perm = (torch.arange(n_img * caps) * 7 + 3) % (n_img * caps)
    batches = []
    for s in range(0, n_img * caps, 16):
        j = perm[s:s + 16]
        batches.append((images[j // caps], tokens[j], iids[j // caps], aids[j], j))
    ev = COCOEvaluator("matmul", n_crossfolds=2, extract_device="cuda", eval_device="cuda")
    ev.set_model(m)
    loader = FakeLoader(batches, FakeDataset(n_img, n_img * caps))
    scores = ev.evaluate(loader, n_images_per_crossfold=6, n_captions_per_crossfold=30, eval_batch_size=32)
    ex = ev.extract_features(loader)
    exn = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in ex.items()}
    exp = ro.evaluate(exn, n_crossfolds=2, n_images_per_crossfold=6, n_captions_per_crossfold=30)
    for task in ("i2t", "t2i"):
        for k, v in exp[task].items():
            assert float(scores[task][k]) == pytest.approx(v, rel=1e-12), (task, k)
        for k, v in exp["n_fold"][task].items():
            assert float(scores["n_fold"][task][k]) == pytest.approx(v, rel=1e-12), (task, k)
    assert float(scores["rsum"]) == pytest.approx(exp["rsum"], rel=1e-12)
    # features are unit-norm rows of the feat_out head
    assert torch.allclose(ex["image_features"].norm(dim=-1), torch.ones(n_img, 1, dtype=torch.float64), atol=1e-5)


def test_retrieve_topk_matches_fp64_matmul_and_stable_sort():
    """COCOEvaluator.retrieve (eval_coco.py:243-288; not on the evaluation path): similarities from the library's fp64 MFMA kernel, the
    reference's stable ascending sort of the negated similarities, top-k gallery ids and scores per query."""
    from fedcola_amd.metrics.eval_coco import COCOEvaluator
    g = torch.Generator().manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(37, 48, generator=g, dtype=torch.float64), dim=1)
    ga = torch.nn.functional.normalize(torch.randn(211, 48, generator=g, dtype=torch.float64), dim=1)
    ga[17] = ga[3]                                   # an exact tie: the stable sort keeps the lower gallery index first
    q_ids, g_ids = list(range(100, 137)), list(range(1000, 1211))
    items, scores, _ = COCOEvaluator(eval_device="cuda").retrieve(q.numpy(), ga.numpy(), q_ids, g_ids, topk=7, batch_size=16)
    sims, pred = (-(q @ ga.t())).sort(stable=True)
    for r, qi in enumerate(q_ids):
        assert items[qi] == [g_ids[j] for j in pred[r, :7].tolist()]
        assert np.allclose(scores[qi], sims[r, :7].numpy(), rtol=0, atol=1e-12)
