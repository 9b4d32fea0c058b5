"""N1 (retrieval evaluation): the CPU oracle against golden vectors produced by the reference's COCOEvaluator
(tests/golden/retrieval.json <- tests/golden/make_golden.py retrieval), and the product's host-side bookkeeping."""
import json
import os

import numpy as np
import pytest

from oracle import retrieval_oracle as ro
from retrieval_util import RETRIEVAL_CASES, FakeDataset, FakeLoader, PassThroughModel, retrieval_set, stream

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "retrieval.json")))


def build(c):
    img, cap, iids, aids = retrieval_set(c["n_images"], c["caps"], c["D"], c["seed"])
    return img, cap, iids, aids, stream(img, cap, iids, aids, c["caps"], c["batch"])


def check_scores(got, exp):
    for k, v in exp.items():
        if isinstance(v, dict):
            check_scores(got[k], v)
        else:
            assert got[k] == pytest.approx(v, rel=1e-12, abs=1e-12), k


@pytest.mark.parametrize("name", list(RETRIEVAL_CASES))
def test_oracle_matches_reference_evaluator(name):
    c, g = RETRIEVAL_CASES[name], GOLD[name]
    assert g["cfg"] == c
    img, cap, iids, aids, batches = build(c)
    ex = ro.collect_features([(b[0].numpy(), b[1].numpy(), b[2].tolist(), b[3].tolist()) for b in batches], c["n_images"], cap.shape[0], c["D"])
    assert [int(v) for v in ex["image_ids"]] == g["image_ids"]
    assert [int(v) for v in ex["caption_ids"]] == g["caption_ids"]
    assert [int(v) for v in ex["caption_classes"]] == g["caption_classes"]
    assert float(ex["image_features"].sum()) == pytest.approx(g["image_feature_sum"], rel=1e-12)
    assert float(ex["caption_features"].sum()) == pytest.approx(g["caption_feature_sum"], rel=1e-12)
    r_i2t = ro.best_ranks(ex["image_features"], ex["caption_features"], ex["image_classes"], ex["caption_classes"])
    r_t2i = ro.best_ranks(ex["caption_features"], ex["image_features"], ex["caption_classes"], ex["image_classes"])
    assert [int(r) for r in r_i2t] == g["ranks_i2t"]           # integer work: bit-exact
    assert [int(r) for r in r_t2i] == g["ranks_t2i"]
    scores = ro.evaluate(ex, n_crossfolds=c["folds"], n_images_per_crossfold=c["ipf"], n_captions_per_crossfold=c["cpf"])
    check_scores(scores, g["scores"])


def test_oracle_tie_rule_and_errors():
    # identical gallery rows tie exactly: the lower gallery index ranks first
    q = np.array([[1.0, 0.0]])
    g = np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    assert ro.best_ranks(q, g, np.array([7]), np.array([3, 7, 7]))[0] == 1
    assert ro.best_ranks(q, g, np.array([7]), np.array([7, 3, 3]))[0] == 0
    with pytest.raises(RuntimeError):
        ro.evaluate_recall(q, g, np.array([7, 8]), np.array([3, 7, 7]))
    with pytest.raises(ValueError):
        ro.best_ranks(q, g, np.array([9]), np.array([3, 7, 7]))     # no positive: min() of an empty list, like the reference


@pytest.mark.parametrize("name", list(RETRIEVAL_CASES))
def test_product_bookkeeping_matches_reference(name):
    """fedcola_amd.metrics.eval_coco.collect (vectorised extract_features bookkeeping) == the reference's per-sample loop."""
    from fedcola_amd.metrics import eval_coco as ec
    c, g = RETRIEVAL_CASES[name], GOLD[name]
    img, cap, iids, aids, batches = build(c)
    import torch
    ex = ec.collect(torch.cat([b[0] for b in batches]), torch.cat([b[1] for b in batches]), torch.cat([b[2] for b in batches]),
                    torch.cat([b[3] for b in batches]), c["n_images"], cap.shape[0], None)
    assert [int(v) for v in ex["image_ids"]] == g["image_ids"]
    assert [int(v) for v in ex["caption_ids"]] == g["caption_ids"]
    assert [int(v) for v in ex["caption_classes"]] == g["caption_classes"]
    assert ex["image_features"].dtype == torch.float64 and tuple(ex["image_features"].shape) == (c["n_images"], 1, c["D"])
    assert float(ex["image_features"].sum()) == pytest.approx(g["image_feature_sum"], rel=1e-12)
    assert float(ex["caption_features"].sum()) == pytest.approx(g["caption_feature_sum"], rel=1e-12)
    ref = ro.collect_features([(b[0].numpy(), b[1].numpy(), b[2].tolist(), b[3].tolist()) for b in batches], c["n_images"], cap.shape[0], c["D"])
    assert np.array_equal(ex["caption_features"].numpy(), ref["caption_features"])
    assert np.array_equal(ex["image_features"].numpy(), ref["image_features"])
    # class map: no regrouping (eval_coco.py:219), classes from the map
    cmap = {int(i): int(i) % 5 for i in iids.tolist()}
    ex2 = ec.collect(torch.cat([b[0] for b in batches]), torch.cat([b[1] for b in batches]), torch.cat([b[2] for b in batches]),
                     torch.cat([b[3] for b in batches]), c["n_images"], cap.shape[0], cmap)
    ref2 = ro.collect_features([(b[0].numpy(), b[1].numpy(), b[2].tolist(), b[3].tolist()) for b in batches], c["n_images"], cap.shape[0], c["D"], cmap)
    for k in ("image_ids", "caption_ids"):
        assert np.array_equal(np.asarray(ex2[k]), ref2[k]), k
    for k in ("image_classes", "caption_classes", "caption_features"):
        assert np.array_equal(ex2[k].numpy(), ref2[k]), k
