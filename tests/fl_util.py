"""Synthetic client datasets and the configuration of the full-round fixture (tests/golden/server_update.json), shared by the generator
(tests/golden/make_golden.py, which runs the REAL reference FedavgServer.update()) and by the tests that drive the product's."""
import torch

from synth import det_ids, det_tensor


class SynthPairs(torch.utils.data.Dataset):
    def __init__(self, n, seq, vocab, seed=0):
        self.img = det_tensor((n, 3, 224, 224), 2000 + seed, 0.5)
        self.ids = det_ids((n, seq), 11 + seed, vocab)

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.ids[i], i // 5, i, i


class SynthCls(torch.utils.data.Dataset):
    def __init__(self, n, kind, classes, seq=8, vocab=30, seed=0):
        self.x = det_tensor((n, 3, 224, 224), 3000 + seed, 0.5) if kind == "img" else det_ids((n, seq), 13 + seed, vocab)
        self.y = (torch.arange(n) * 7 + seed) % classes

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


# ---- full-round fixture: 3 rounds of FedavgServer.update() with aux refresh, warm-up filter, freeze / unfreeze and LR decay
ROUND_DS = {"CIFAR100": ("cls", "img"), "AG_NEWS": ("cls", "txt"), "Flickr30k": ("rtv", "img+txt")}
ROUND_COMMON = dict(embed_dim=4, depth=1, num_heads=2, vocab_size=30, max_text_len=8)
ROUND_ARGS = dict(shared_param="attn", share_scope="all", compensation=True, with_aux=True, aux_trained=False,
                  datasets=list(ROUND_DS.keys()), modalities=["img", "txt", "img+txt"], out_modality_scales=[1, 0.5, 1],
                  E=1, B=4, lr=1e-3, optimizer="AdamW", no_shuffle=True, equal_sampled=True, K=6,
                  freeze_modality="img", freeze_rounds=1, warmup_modality="txt", warmup_rounds=1, lr_decay=0.9, lr_decay_step=1,
                  num_thread=1, seq_len=8, vocab_size=30)
ROUND_CS = {"CIFAR100": 1.0, "AG_NEWS": 0.5, "Flickr30k": 0.5}
ROUND_SEED = 5
ROUND_N = 3
# (client id, dataset name, number of training samples)
ROUND_LAYOUT = [(0, "CIFAR100", 8), (1, "CIFAR100", 6), (2, "AG_NEWS", 8), (3, "AG_NEWS", 7), (4, "Flickr30k", 8), (5, "Flickr30k", 5)]


def round_model_kwargs(ds):
    task, mod = ROUND_DS[ds]
    if mod == "img":
        return dict(modalities=["img", None], num_classes=[100, None], tasks=["cls", None], **ROUND_COMMON)
    if mod == "txt":
        return dict(modalities=[None, "txt"], num_classes=[None, 4], tasks=[None, "cls"], **ROUND_COMMON)
    return dict(modalities=["img", "txt"], num_classes=[None, None], tasks=["rtv", "rtv"], **ROUND_COMMON)


def round_dataset(cid, ds, n):
    mod = ROUND_DS[ds][1]
    if mod == "img":
        return SynthCls(n, "img", 100, seed=cid)
    if mod == "txt":
        return SynthCls(n, "txt", 4, seq=8, vocab=30, seed=cid)
    return SynthPairs(n, 8, 30, seed=cid)
