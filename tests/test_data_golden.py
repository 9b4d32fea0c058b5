"""N4 (input side of the path): client splits bit-exact against the reference's numpy RNG stream (tests/golden/split.json),
the Flickr30k on-disk format, and the device prefetcher's ordering contract (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from refstub import RefArgs, reference_available

HERE = os.path.dirname(os.path.abspath(__file__))
SPLITS = json.load(open(os.path.join(HERE, "golden", "split.json")))


class _DS:
    def __init__(self, n, ncls):
        self.targets = [(i * 7 + i // 3) % ncls for i in range(n)]

    def __len__(self):
        return len(self.targets)


@pytest.mark.parametrize("rec", SPLITS, ids=[r["cfg"]["name"] for r in SPLITS])
def test_simulate_split_bit_exact(rec):
    from fedcola_amd.loaders.split import simulate_split
    c = rec["cfg"]
    a = RefArgs(**{k: v for k, v in c.items() if k not in ("name", "n", "seed")})
    np.random.seed(c["seed"])
    m = simulate_split(a, _DS(c["n"], c.get("num_classes", 10)))
    assert {str(k): [int(x) for x in v] for k, v in m.items()} == rec["map"]           # index work: bit-exact
    assert float(np.random.uniform()) == rec["next_uniform"]                             # same number of RNG draws consumed
    if c["dataset"] in ("Flickr30k", "Coco"):
        for v in m.values():                                                             # whole images: 5 consecutive captions
            v = np.asarray(v)
            assert len(v) % 5 == 0 and np.all(v[::5] % 5 == 0) and np.all(np.diff(v.reshape(-1, 5), axis=1) == 1)


def test_split_errors():
    from fedcola_amd.loaders.split import simulate_split
    with pytest.raises(AssertionError):
        simulate_split(RefArgs(split_type="patho", dataset="CIFAR100", K=4, mincls=1, num_classes=10), _DS(40, 10))
    with pytest.raises(Exception):
        simulate_split(RefArgs(split_type="patho", dataset="CIFAR100", K=2, mincls=2, num_classes=10), _DS(40, 10))


def _make_flickr(root, n_images=3):
    from PIL import Image
    os.makedirs(os.path.join(root, "flickr30k_images"), exist_ok=True)
    rows = ["image_name| comment_number| comment"]
    for i in range(n_images):
        name = f"{1000 + i}.jpg"
        Image.fromarray((np.arange(8 * 8 * 3, dtype=np.uint8).reshape(8, 8, 3) + 13 * i)).save(os.path.join(root, "flickr30k_images", name.replace(".jpg", ".png")))
        os.replace(os.path.join(root, "flickr30k_images", name.replace(".jpg", ".png")), os.path.join(root, "flickr30k_images", name))
        for j in range(5):
            rows.append(f"{name}| {j}| A caption number {j} , with a comma and \"quotes\" for image {i} .")
    for split in ("train", "test", "train_all"):
        with open(os.path.join(root, f"{split}.csv"), "w") as f:
            f.write("\n".join(rows if split != "test" else rows[:6]) + "\n")


def _tok(text, padding=None, truncation=None, max_length=8, return_tensors=None):
    ids = [min(len(w), 29) for w in text.split()][:max_length]
    return {"input_ids": torch.tensor([ids + [0] * (max_length - len(ids))])}


def test_flickr30k_format(tmp_path):
    from fedcola_amd.datasets.flickr30k import Flickr30kCap, fetch_flickr30k
    root = str(tmp_path)
    _make_flickr(root)
    to_t = lambda im: torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1)
    ds = Flickr30kCap(root, split="train", transform=to_t, tokenizer=_tok, max_length=8)
    assert len(ds) == 15 and ds.n_images == 3 and ds.iid_to_cls == {}
    img, cap, iid, aid, idx = ds[7]
    assert (iid, aid, idx) == (1, 7, 7) and tuple(img.shape) == (3, 8, 8) and cap.shape == (8,)
    assert ds.captions[7].strip().startswith("A caption number 2 , with a comma")
    a = RefArgs(seq_len=8, flickr_train_all=True)
    tr, te, a2 = fetch_flickr30k(a, root, (to_t, to_t), _tok)
    assert len(tr) == 15 and len(te) == 5 and tr.name == "Flickr30k" and tr.task == "img+txt" and a2.in_channels == 3 and a2.num_classes is None
    if reference_available():          # same parse as the reference's class, sample by sample
        import refstub
        refstub.load_reference()
        import importlib
        ref = importlib.import_module("src.datasets.flickr30k")
        rd = ref.Flickr30kCap(root, split="train", transform=to_t, tokenizer=_tok, max_length=8)
        assert rd.captions == ds.captions and list(rd.images) == list(ds.images) and rd.n_images == ds.n_images
        for i in (0, 4, 5, 14):
            x, y = rd[i], ds[i]
            assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and x[2:] == y[2:]


@pytest.mark.gpu
def test_device_prefetcher_order_and_values():
    from fedcola_amd.loaders.prefetch import DevicePrefetcher
    batches = [(torch.full((4, 3, 16, 16), float(i)), torch.arange(8).repeat(4, 1) + i, i) for i in range(7)]
    for depth in (1, 2, 3):
        got = list(DevicePrefetcher(batches, "cuda", depth=depth))
        assert len(got) == 7
        for i, (x, y, k) in enumerate(got):
            assert x.is_cuda and y.is_cuda and k == i
            assert float(x.mean()) == float(i) and int(y[0, 0]) == i
    with pytest.raises(RuntimeError):
        list(DevicePrefetcher(batches, "cpu"))


def _make_coco(root, n_images=4, caps=5):
    from PIL import Image
    os.makedirs(os.path.join(root, "all_images"), exist_ok=True)
    os.makedirs(os.path.join(root, "annotations"), exist_ok=True)
    for split, base in (("train", 100), ("val", 500)):
        images, anns = [], []
        for i in range(n_images):
            iid = base + 7 * i
            fn = f"COCO_{split}2014_{iid:012d}.jpg"
            Image.fromarray(((np.arange(8 * 8 * 3).reshape(8, 8, 3) + iid) % 255).astype(np.uint8)).save(os.path.join(root, "all_images", fn.replace(".jpg", ".png")))
            os.replace(os.path.join(root, "all_images", fn.replace(".jpg", ".png")), os.path.join(root, "all_images", fn))
            images.append({"id": iid, "file_name": fn})
            for j in range(caps):
                anns.append({"id": 10 * iid + j, "image_id": iid, "caption": f"caption {j} of image {iid}"})
        with open(os.path.join(root, "annotations", f"captions_{split}2014.json"), "w") as f:
            json.dump({"info": {}, "images": images, "annotations": anns}, f)
        np.save(os.path.join(root, f"coco_{'train' if split == 'train' else 'test'}_ids.npy"), np.array([a["id"] for a in anns], dtype=np.int64))
    os.makedirs(os.path.join(root, "inst"), exist_ok=True)
    with open(os.path.join(root, "inst", "instances_train2014.json"), "w") as f:
        json.dump({"annotations": [{"image_id": 100, "category_id": 3}, {"image_id": 107, "category_id": 3}, {"image_id": 114, "category_id": 90},
                                   {"image_id": 121, "category_id": 1}, {"image_id": 121, "category_id": 3}]}, f)


def test_coco_format(tmp_path):
    """COCO caption annotations without pycocotools (parity unpinned there: the published annotation format is the contract)."""
    from fedcola_amd.datasets.coco import CocoCaptionsCap, fetch_coco, public_set
    root = str(tmp_path)
    _make_coco(root)
    to_t = lambda im: torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1)
    ann = os.path.join(root, "annotations", "captions_train2014.json")
    ds = CocoCaptionsCap(os.path.join(root, "all_images"), ann, transform=to_t, tokenizer=_tok, max_length=8)
    assert len(ds) == 20 and ds.n_images == 4 and ds.iid_to_cls == {}
    img, cap, iid, aid, idx = ds[7]
    assert (iid, aid, idx) == (107, 1072, 7) and tuple(img.shape) == (3, 8, 8) and cap.shape == (8,)
    sub = CocoCaptionsCap(os.path.join(root, "all_images"), ann, ids=np.array([1000, 1141, 1142]), extra_ids=[1210])
    assert sub.ids == [1000, 1141, 1142, 1210] and sub.n_images == 3 and sub[1][1] == "caption 1 of image 114"
    sub.reduce_samples(2)
    assert sub.ids == [1142, 1210]
    cls = CocoCaptionsCap(os.path.join(root, "all_images"), ann, instance_annFile=os.path.join(root, "inst"))
    assert cls.iid_to_cls == {100: 0, 107: 0, 114: 1, 121: 2}          # same category code -> same dense class, first-seen order
    merged = CocoCaptionsCap(os.path.join(root, "all_images"), ann, extra_annFile=os.path.join(root, "annotations", "captions_val2014.json"))
    assert len(merged) == 40 and merged.n_images == 8
    a = RefArgs(seq_len=8, reduce_samples=10)
    tr, te, a2 = fetch_coco(a, root, (to_t, to_t), _tok)
    assert len(tr) == 10 and len(te) == 20 and tr.name == "Coco" and a2.num_classes is None
    pub = public_set(os.path.join(root, "all_images"), root + "/annotations/captions_train2014.json", 6, transform=to_t, tokenizer=_tok, max_length=8)
    assert pub.ids == [1144, 1210, 1211, 1212, 1213, 1214]          # the LAST ids of coco_train_ids.npy


def test_pinned_batch_loader_yields_the_dataloaders_batches():
    """Same sampler, same RNG state -> the same batches as torch's DataLoader (shuffled and sequential, ragged last batch, mixed fields)."""
    from fedcola_amd.loaders.batch import PinnedBatchLoader

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 23

        def __getitem__(self, i):
            return torch.full((3, 5, 5), float(i)), torch.arange(4) + i, i // 5, i, i

    for shuffle in (False, True):
        torch.manual_seed(11)
        ref = list(torch.utils.data.DataLoader(DS(), batch_size=6, shuffle=shuffle))
        torch.manual_seed(11)
        got = list(PinnedBatchLoader(DS(), 6, shuffle=shuffle, workers=4))
        assert len(got) == len(ref) == 4 and len(PinnedBatchLoader(DS(), 6)) == 4
        for a, b in zip(got, ref):
            assert len(a) == len(b) == 5
            for x, y in zip(a, b):
                assert x.dtype == y.dtype and torch.equal(x, y)
    assert len(list(PinnedBatchLoader(DS(), 6, drop_last=True))) == 3

    class DSB(DS):
        def get_batch(self, idxs):
            i = torch.as_tensor(idxs)
            return torch.stack([torch.full((3, 5, 5), float(k)) for k in idxs]), torch.arange(4)[None, :] + i[:, None], i // 5, i, i
    torch.manual_seed(11)
    ref = list(torch.utils.data.DataLoader(DS(), batch_size=6, shuffle=True))
    torch.manual_seed(11)
    for a, b in zip(PinnedBatchLoader(DSB(), 6, shuffle=True, ahead=0), ref):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and torch.equal(x, y)


def test_batch_iterator_draws_its_rng_numbers_when_created_and_can_be_abandoned():
    """The epoch iterator is an object: the sampler's RNG draws happen when iter() is CALLED (FedavgClient.update() creates it before it
    allocates the optimizer state so that the first batch is assembled meanwhile), not at the first next(); an abandoned epoch ends its
    producer thread on close(); an exhausted iterator stays exhausted."""
    import threading
    import time
    from fedcola_amd.loaders.batch import PinnedBatchLoader

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 23

        def __getitem__(self, i):
            return torch.full((2, 3), float(i)), i

    torch.manual_seed(11)
    ref = list(torch.utils.data.DataLoader(DS(), batch_size=6, shuffle=True))
    ld = PinnedBatchLoader(DS(), 6, shuffle=True, workers=2)
    torch.manual_seed(11)
    it = iter(ld)
    torch.manual_seed(999)                                   # whatever the caller does to the generator afterwards changes nothing
    got = list(it)
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))
    with pytest.raises(StopIteration):
        next(it)
    base = threading.active_count()
    it2 = iter(ld)
    next(it2)
    assert threading.active_count() == base + 1              # the producer of the open epoch
    it2.close()
    for _ in range(50):
        if threading.active_count() == base:
            break
        time.sleep(0.02)
    assert threading.active_count() == base


# ---------------------------------------------------------------------------------------------- tokenizer (data.py:182-190)
def test_bert_vocab_tokenizer_matches_hf_golden_ids():
    import json
    from fedcola_amd.loaders.tokenizer import BertVocabTokenizer
    voc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flickr30k_vocab.txt")
    tok = BertVocabTokenizer(voc)
    assert tok.vocab_size == 7732                                   # VOCAB_SIZES['Flickr30k'], fedavgserver.py:89-92
    assert (tok.pad_token_id, tok.unk_token_id, tok.cls_token_id, tok.sep_token_id) == (0, 100, 101, 102)
    recs = json.load(open(os.path.join(os.path.dirname(voc), "tokenizer.json")))
    assert len(recs) >= 200
    for r in recs:
        got = tok(r["text"], padding="max_length", truncation=True, max_length=r["max_length"], return_tensors="pt")["input_ids"][0]
        assert got.dtype == torch.int64 and got.tolist() == r["ids"], r["text"][:60]
    # the partial(tokenizer, padding='max_length', max_length=seq_len, truncation=True) form of data.py:299-303
    out = tok("a man in a blue shirt", padding="max_length", max_length=12, truncation=True)
    assert len(out["input_ids"]) == 12 and out["attention_mask"] == [1] * 8 + [0] * 4      # [CLS] + 6 words + [SEP], then 4 x [PAD]
    assert out["input_ids"][0] == 101 and out["input_ids"][7] == 102 and out["input_ids"][8:] == [0] * 4
    with pytest.raises(ValueError):
        BertVocabTokenizer("/nonexistent/vocab.txt")


def test_bert_vocab_tokenizer_against_live_hf_tokenizer():
    """Where transformers is installed (it is in this image): random sentences beyond the golden set."""
    transformers = pytest.importorskip("transformers")
    import random
    from fedcola_amd.loaders.tokenizer import BertVocabTokenizer
    voc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flickr30k_vocab.txt")
    hf, mine = transformers.BertTokenizer(voc), BertVocabTokenizer(voc)
    words = [l.rstrip("\n") for l in open(voc)]
    rng = random.Random(123)
    for _ in range(200):
        s = " ".join(rng.choice(words) if rng.random() > 0.1 else rng.choice(["Xyzzy", "re-do", "it's", "über", "A.B.C", "12:30pm"]) for _ in range(rng.randint(0, 50)))
        for L in (32, 40):
            assert mine(s, padding="max_length", truncation=True, max_length=L, return_tensors="pt")["input_ids"][0].tolist() == \
                hf(s, padding="max_length", truncation=True, max_length=L, return_tensors="pt")["input_ids"][0].tolist(), s


def _imnorm(size):
    """The reference's --resize N --imnorm chain (Resize, ToTensor, Normalize(0.5, 0.5); src/loaders/data.py:90-109) without
    torchvision: PIL resize + the tensor ops of torchvision's to_tensor / normalize."""
    def t(im):
        im = im.resize((size, size))
        x = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)
        return x.sub_(0.5).div_(0.5)
    return t


def test_decoded_cache_serves_the_datasets_own_samples_bit_for_bit(tmp_path):
    """loaders.cache.DecodedCache over Flickr30kCap (also behind a Subset + the reference's SubsetWrapper): every sample and every batch
    equals the dataset's own, bit for bit; the store is uint8 (lossless by verification) and holds one image per five captions; the
    client's loader over it yields the DataLoader's batches in the DataLoader's order under the same RNG state."""
    from fedcola_amd.datasets.flickr30k import Flickr30kCap
    from fedcola_amd.loaders import DecodedCache, PinnedBatchLoader
    root = str(tmp_path)
    _make_flickr(root, n_images=7)
    ds = Flickr30kCap(root, split="train", transform=_imnorm(12), tokenizer=_tok, max_length=8)
    assert DecodedCache.applicable(ds)
    dc = DecodedCache(ds).build()
    assert dc.lut is not None and dc.u8.shape == (7, 3, 12, 12) and len(dc) == 35
    for i in (0, 4, 5, 17, 34):
        a, b = ds[i], dc[i]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2:] == b[2:]
    got = dc.get_batch([3, 33, 10])
    for j, i in enumerate([3, 33, 10]):
        assert torch.equal(got[0][j], ds[i][0]) and torch.equal(got[1][j], ds[i][1]) and [int(g[j]) for g in got[2:]] == list(ds[i][2:])

    class SubsetWrapper(torch.utils.data.Dataset):          # src/loaders/data.py:35-49
        def __init__(self, subset):
            self.subset = subset
        def __getitem__(self, i):
            return self.subset[i]
        def __len__(self):
            return len(self.subset)
    sub = SubsetWrapper(torch.utils.data.Subset(ds, [30, 31, 32, 33, 34, 5, 6, 7, 8, 9, 20, 21]))
    assert DecodedCache.applicable(sub)
    sc = DecodedCache(sub)
    torch.manual_seed(11)
    ref = list(torch.utils.data.DataLoader(sub, batch_size=5, shuffle=True))
    torch.manual_seed(11)
    mine = list(PinnedBatchLoader(sc, 5, shuffle=True, workers=3, pin=False))
    assert sc.built and sc.u8.shape[0] == 3                 # three images behind the twelve caption samples
    assert len(ref) == len(mine) == 3
    for r, m in zip(ref, mine):
        assert len(r) == len(m)
        for x, y in zip(r, m):
            assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))


def test_decoded_cache_refuses_random_transforms_and_keeps_floats_when_uint8_would_lose_bits(tmp_path):
    from fedcola_amd.datasets.flickr30k import Flickr30kCap
    from fedcola_amd.loaders import DecodedCache
    root = str(tmp_path)
    _make_flickr(root, n_images=3)
    base = _imnorm(8)
    rnd = Flickr30kCap(root, split="train", transform=lambda im: base(im) + torch.rand(()) * 1e-3, tokenizer=_tok, max_length=8)
    assert not DecodedCache.applicable(rnd)                 # a random transform: the client keeps the reference's DataLoader
    odd = Flickr30kCap(root, split="train", transform=lambda im: base(im) * 1.0001, tokenizer=_tok, max_length=8)
    dc = DecodedCache(odd).build()
    assert dc.lut is None and dc.f32.shape == (3, 3, 8, 8)  # not a ToTensor / Normalize table: the floats themselves are kept
    assert torch.equal(dc[7][0], odd[7][0]) and torch.equal(dc.get_batch([7, 1])[0][0], odd[7][0])
    with pytest.raises(ValueError):
        DecodedCache(odd, store="uint8").build()
    assert not DecodedCache.applicable(torch.utils.data.TensorDataset(torch.zeros(4, 3, 8, 8)))     # no image_key: not a caption dataset


def test_decoded_cache_applicability_is_decided_from_the_transform_chain(tmp_path):
    """one pair of equal fetches does not prove a transform deterministic -- RandomHorizontalFlip(0.5) passes it every second
    time and the cache would then freeze ONE augmented view per image.  Random* / ColorJitter members refuse by class name without a fetch;
    chains of known deterministic transforms accept without a fetch; unknown callables are probed several times on several samples, and no
    probe may move the torch / numpy / python RNG streams (the run's shuffle and augmentation order must stay the reference's)."""
    import random
    from fedcola_amd.datasets.flickr30k import Flickr30kCap
    from fedcola_amd.loaders import DecodedCache
    root = str(tmp_path)
    _make_flickr(root, n_images=4)
    base = _imnorm(8)

    class RandomHorizontalFlip:                      # torchvision's class name and behaviour (p = 0.5, one torch.rand(1) draw per call)
        def __call__(self, x):
            return x.flip(-1) if torch.rand(1) < 0.5 else x

    class ToTensor:                                  # a known deterministic name
        def __call__(self, im):
            return base(im)

    class Compose:
        def __init__(self, ts):
            self.transforms = ts
        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    calls = []

    class Counting(Flickr30kCap):
        def __getitem__(self, i):
            calls.append(i)
            return super().__getitem__(i)

    def states():
        return torch.get_rng_state().clone(), np.random.get_state()[1].copy(), random.getstate()

    def same(a, b):
        return torch.equal(a[0], b[0]) and bool((a[1] == b[1]).all()) and a[2] == b[2]

    # (1) a named random member: refused without fetching a sample or drawing a number
    ds = Counting(root, split="train", transform=Compose([ToTensor(), RandomHorizontalFlip()]), tokenizer=_tok, max_length=8)
    s0 = states()
    assert not DecodedCache.applicable(ds) and calls == [] and same(s0, states())
    # (2) the same flip hidden in a lambda (unknown by name): the repeated probe catches it, and the RNG streams come back untouched
    flip = RandomHorizontalFlip()
    hidden = Counting(root, split="train", transform=lambda im: flip(base(im)), tokenizer=_tok, max_length=8)
    verdicts = []
    for seed in range(20):
        torch.manual_seed(seed)
        s0 = states()
        verdicts.append(DecodedCache.applicable(hidden))
        assert same(s0, states())
    assert sum(verdicts) <= 1                        # 3 samples x 4 fetches: all agree with probability 2^-9 per call
    # (3) known deterministic names only: still probed, lightly -- ONE sample fetched twice (randomness can sit outside `transform`)
    calls.clear()
    det = Counting(root, split="train", transform=Compose([ToTensor()]), tokenizer=_tok, max_length=8)
    assert DecodedCache.applicable(det) and calls == [0, 0]
    # (3b) ... which is what catches a dataset whose CAPTION choice is random although its transform chain is all deterministic names
    class RandomCaption(Flickr30kCap):
        def __getitem__(self, i):
            s = list(super().__getitem__(i))
            s[1] = s[1].clone()
            s[1][1] = int(torch.randint(5, 25, (1,)))
            return tuple(s)
    rc = RandomCaption(root, split="train", transform=Compose([ToTensor()]), tokenizer=_tok, max_length=8)
    s0 = states()
    assert sum(DecodedCache.applicable(rc) for _ in range(10)) <= 2 and same(s0, states())
    # (4) an unknown but deterministic callable (the reference pads its chains with identity Lambdas): accepted after the probe
    calls.clear()
    lam = Counting(root, split="train", transform=lambda im: base(im), tokenizer=_tok, max_length=8)
    assert DecodedCache.applicable(lam) and len(calls) == 3 * 4
