#!/usr/bin/env python
"""Host batch assembly: torch's single-process DataLoader (the reference's loader) vs fedcola_amd.loaders.PinnedBatchLoader on a
Flickr30k-shaped tensor dataset (B = 64, 3x224x224 fp32 + 32 tokens).  usage: tools/loader_bench.py [n_samples=1024]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedcola_amd.loaders import PinnedBatchLoader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


class DS(torch.utils.data.Dataset):
    def __init__(self):
        self.img = torch.randn(n, 3, 224, 224)
        self.ids = torch.randint(0, 7732, (n, 32))

    def __len__(self):
        return n

    def __getitem__(self, i):
        return self.img[i], self.ids[i], i // 5, i, i


class DSB(DS):
    def get_batch(self, idxs):
        i = torch.as_tensor(idxs)
        return self.img[i], self.ids[i], i // 5, i, i


ds = DS()
dsb = DSB()
out = {}
for name, mk in (("DataLoader", lambda: torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True)),
                 ("PinnedBatchLoader", lambda: PinnedBatchLoader(ds, 64, shuffle=True, workers=4)),
                 ("PinnedBatchLoader+get_batch", lambda: PinnedBatchLoader(dsb, 64, shuffle=True, workers=4))):
    list(zip(range(2), mk()))
    t0 = time.perf_counter()
    nb = sum(1 for _ in mk())
    out[name] = round((time.perf_counter() - t0) / nb * 1e3, 2)
print(json.dumps(dict(ms_per_batch=out, batch="64 x (3x224x224 fp32 + 32 int64 tokens)", host_threads=torch.get_num_threads())))
# end to end: the ViT-S client step fed by each loader through DevicePrefetcher (host batches -> H2D on the library's side stream)
if torch.cuda.is_available():
    from bench import Args
    from fedcola_amd import _lib
    from fedcola_amd.loaders import DevicePrefetcher
    from fedcola_amd.mome import create_model
    a = Args(); a.precision = "bf16"
    dev = torch.device("cuda")
    model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
    model.train()
    nn_ = model.flat.numel()
    grads = torch.zeros(nn_, device=dev); m1 = torch.zeros(nn_, device=dev); m2 = torch.zeros(nn_, device=dev); lossbuf = torch.zeros(2, device=dev)
    model.prepare_weights(force=True); ws = model.workspace(64, 32)
    L, P = _lib.lib(), _lib.ptr
    e2e = {}
    for name, mk in (("DataLoader", lambda: torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True, drop_last=True)),
                     ("PinnedBatchLoader", lambda: PinnedBatchLoader(ds, 64, shuffle=True, drop_last=True, workers=4, ahead=2)),
                     ("PinnedBatchLoader+get_batch", lambda: PinnedBatchLoader(dsb, 64, shuffle=True, drop_last=True, workers=4, ahead=2))):
        k = 0
        t0 = None
        for ep in range(2):
            for img, ids, *_ in DevicePrefetcher(mk(), dev, depth=2, stream=model.side_stream()):
                k += 1
                _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img.float()), P(ids), None, 64, 32,
                                            None, 1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lossbuf), P(ws), ws.numel(), _lib.stream_ptr()))
            if ep == 0:
                torch.cuda.synchronize(); t0 = time.perf_counter(); k0 = k
        torch.cuda.synchronize()
        e2e[name] = round((time.perf_counter() - t0) / (k - k0) * 1e3, 2)
    print(json.dumps(dict(client_loop_ms_per_step=e2e, note="ViT-S bf16 fused step fed from host memory; the device step alone is bench.py's ms_per_step (4.8 ms in round 3)")))

# ---- the REAL dataset class: Flickr30kCap over JPEG files on disk (synthetic 500x375 photos-sized noise images, five captions each),
# the reference's --resize 224 --imnorm transform and the Flickr30k vocabulary: reference DataLoader vs the client's default loader
# (DecodedCache: decoded once per client, uint8 store) -- host side, then the client loop on the device.
import tempfile
import numpy as np
from PIL import Image
from fedcola_amd.datasets.flickr30k import Flickr30kCap
from fedcola_amd.loaders import DecodedCache
from fedcola_amd.loaders.tokenizer import BertVocabTokenizer
n_img = int(os.environ.get("LOADER_BENCH_IMAGES", "256"))
root = tempfile.mkdtemp(prefix="fc_flickr_")
os.makedirs(os.path.join(root, "flickr30k_images"))
rng = np.random.default_rng(0)
rows = ["image_name| comment_number| comment"]
words = "a man in a red shirt is riding a bike down the street while two dogs play on the grass near a small white house".split()
for i in range(n_img):
    name = f"{100000 + i}.jpg"
    Image.fromarray(rng.integers(0, 255, (375, 500, 3), dtype=np.uint8)).save(os.path.join(root, "flickr30k_images", name), quality=90)
    for j in range(5):
        rows.append(f"{name}| {j}| " + " ".join(rng.choice(words, 12)) + " .")
for split in ("train", "test"):
    open(os.path.join(root, f"{split}.csv"), "w").write("\n".join(rows) + "\n")
voc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "flickr30k_vocab.txt")


def tf(im):
    im = im.resize((224, 224))
    return torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255).sub_(0.5).div_(0.5)


fds = Flickr30kCap(root, split="train", transform=tf, tokenizer=BertVocabTokenizer(voc), max_length=32)
t0 = time.perf_counter()
ref_batches = 0
for b in torch.utils.data.DataLoader(fds, batch_size=64, shuffle=True):
    ref_batches += 1
    if ref_batches == 6:
        break
ref_ms = (time.perf_counter() - t0) / ref_batches * 1e3
t0 = time.perf_counter()
dc = DecodedCache(fds, workers=8).build()
build_s = time.perf_counter() - t0
def host_ms(raw):
    ld = PinnedBatchLoader(dc, 64, shuffle=True, workers=8, raw=raw)
    list(zip(range(2), ld))
    t0 = time.perf_counter()
    nb = sum(1 for _ in ld)
    return (time.perf_counter() - t0) / nb * 1e3
fast_ms, raw_ms = host_ms(False), host_ms(True)
rec = dict(dataset=f"Flickr30kCap over {n_img} JPEGs (500x375) x 5 captions, resize 224 + imnorm, 32 tokens", reference_DataLoader_ms_per_batch=round(ref_ms, 1),
           decoded_cache_build_s=round(build_s, 2), decoded_cache_store="uint8" if dc.lut is not None else "float32",
           decoded_cache_MB=round((dc.u8.numel() if dc.lut is not None else dc.f32.numel() * 4) / 1e6, 1), PinnedBatchLoader_over_cache_ms_per_batch=round(fast_ms, 2), PinnedBatchLoader_over_cache_uint8_ms_per_batch=round(raw_ms, 2))
if torch.cuda.is_available():
    k = 0
    for ep in range(3):
        for img, ids, *_ in DevicePrefetcher(PinnedBatchLoader(dc, 64, shuffle=True, drop_last=True, workers=8, ahead=2, raw=True), dev, depth=2, stream=model.side_stream()):
            k += 1
            _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img.float()), P(ids), None, 64, 32,
                                        None, 1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lossbuf), P(ws), ws.numel(), _lib.stream_ptr()))
        if ep == 0:
            torch.cuda.synchronize(); t0 = time.perf_counter(); k0 = k
    torch.cuda.synchronize()
    rec["client_loop_ms_per_step_over_cache_uint8_h2d"] = round((time.perf_counter() - t0) / (k - k0) * 1e3, 2)
print(json.dumps(rec))
import shutil
shutil.rmtree(root, ignore_errors=True)
