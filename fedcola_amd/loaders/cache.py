"""Pre-decoded view of an image-caption dataset (VERDICT r03 task 6).

The reference feeds its clients from ``Flickr30kCap`` / ``CocoCaptionsCap`` through a single-process ``DataLoader``
(/root/reference/src/datasets/flickr30k.py:30-42, src/client/fedavgclient.py:44-53): every sample of every epoch is a JPEG decode, a
resize, a float conversion and a tokenizer call -- ~2.4 ms per sample, 157 ms per B = 64 batch on the MI355X host against a 4.5-ms
device step.  With the reference's own transform chain (``--resize 224 --imnorm``: Resize, ToTensor, Normalize -- no random
augmentation; src/loaders/data.py:85-110, scripts/flickr.sh) a sample is a pure function of its index, so it can be computed ONCE
per client and served from memory ever after.

``DecodedCache(dataset)`` wraps a caption dataset (or a ``Subset`` / ``SubsetWrapper`` chain over one) and offers the same
``__getitem__`` tuples plus ``get_batch(indices, out=)`` -- the vectorised fetch ``PinnedBatchLoader`` gathers straight into a
pinned batch.  What it stores:

* images per IMAGE, not per caption sample (five captions share one): ``dataset.image_key(i)`` names the image of sample i;
* as uint8 when that is provably lossless: a candidate per-channel table ``lut[c][u] = Normalize(ToTensor(u))`` (the torch op
  sequence of torchvision's ToTensor / Normalize: ``u.float().div(255)`` then ``.sub(mean).div(std)``) is inverted on every decoded
  image and the image is re-generated from the table -- only if EVERY image comes back bit-identical the uint8 store is used
  (4x smaller than fp32: 150 KB per 224x224 image); otherwise the float tensors themselves are kept (``store="float32"``);
* token ids, image ids, annotation ids per sample.

The build decodes each image once, through the dataset's own ``__getitem__`` (same PIL / transform / tokenizer code), on a few
threads, the first time the cache is used (``build()``; clients persist across rounds, so once per client).  A dataset whose
transform is RANDOM is refused -- decided from the transform chain's class names, with a repeated-fetch probe (RNG state saved and
restored) only for members that cannot be judged by name: ``applicable()`` is False and the client keeps the reference's DataLoader.  Index order is the loader's business (``PinnedBatchLoader`` = the DataLoader's order under the same RNG
state); the values are the dataset's own, bit for bit (tests/test_data_golden.py)."""
from __future__ import annotations

import logging
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from torch.utils.data import Dataset

logger = logging.getLogger(__name__)

# (mean, std) candidates of the ToTensor -> Normalize chain: none (ToTensor only), the reference's --imnorm (0.5 / 0.5)
_NORMS = [None, ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))]


def _resolve(dataset):
    """(base dataset, index map or None) through Subset / SubsetWrapper chains (``.dataset`` + ``.indices``, ``.subset``)."""
    idx = None
    d = dataset
    for _ in range(8):
        if hasattr(d, "image_key"):
            return d, idx
        if hasattr(d, "subset"):                       # the reference's SubsetWrapper (src/loaders/data.py:35-49)
            d = d.subset
        elif hasattr(d, "dataset") and hasattr(d, "indices"):
            ind = np.asarray(d.indices, dtype=np.int64)
            idx = ind if idx is None else ind[idx]
            d = d.dataset
        else:
            break
    return (d, idx) if hasattr(d, "image_key") else (None, None)


def _lut(norm):
    u = torch.arange(256, dtype=torch.uint8)
    x = u.float().div(255)                             # torchvision.transforms.functional.to_tensor
    if norm is None:
        return x[None].repeat(3, 1)
    mean, std = norm
    return torch.stack([x.clone().sub_(m).div_(s) for m, s in zip(mean, std)])      # F.normalize: tensor.sub_(mean).div_(std)


class DecodedCache(Dataset):
    def __init__(self, dataset, store: str = "auto", workers: int = 8, norm=None):
        """store: "auto" (uint8 when lossless, else float32) | "uint8" (raise when not lossless) | "float32".
        norm: an extra (mean, std) candidate for the uint8 table (a dataset-specific Normalize)."""
        self.dataset = dataset
        self.base, self.index = _resolve(dataset)
        if self.base is None:
            raise TypeError("DecodedCache needs a dataset (or Subset chain over one) that offers image_key(index)")
        self.store_kind, self.workers = store, max(1, int(workers))
        self.norms = _NORMS + ([norm] if norm is not None else [])
        self.built = False
        self._lut_dev = {}
        self._lock = threading.Lock()
        for attr in ("task", "modality", "name", "iid_to_cls", "n_images"):      # what the server / evaluator read off a dataset
            if hasattr(dataset, attr):
                setattr(self, attr, getattr(dataset, attr))

    # ------------------------------------------------------------------ applicability
    # transforms by class name (torchvision's; the stub classes of the tests carry the same names): a chain made of the first set only is a
    # pure function of the image, one member of the second set makes the sample random
    _DETERMINISTIC = {"ToTensor", "Normalize", "Resize", "CenterCrop", "ToPILImage", "PILToTensor", "ConvertImageDtype", "Grayscale", "Pad",
                      "FiveCrop", "TenCrop", "Compose"}
    _RANDOM_PREFIXES = ("Random", "Rand", "Auto", "Trivial", "AugMix", "ColorJitter", "GaussianBlur", "ElasticTransform")

    @classmethod
    def _transform_verdict(cls, t):
        """"det" | "random" | "unknown" for a transform (chains through .transforms)."""
        if t is None:
            return "det"
        name = type(t).__name__
        if name.startswith(cls._RANDOM_PREFIXES):
            return "random"
        inner = getattr(t, "transforms", None)
        if inner is not None and not callable(inner):
            verdicts = [cls._transform_verdict(u) for u in inner]
            return "random" if "random" in verdicts else ("unknown" if "unknown" in verdicts else "det")
        return "det" if name in cls._DETERMINISTIC else "unknown"      # Lambda, user callables: cannot be judged by name

    @classmethod
    def applicable(cls, dataset, probe_samples: int = 3, probe_repeats: int = 4) -> bool:
        """A caption dataset (image_key) whose samples are a pure function of the index.  Decided from the transform chain: any Random* /
        ColorJitter / Auto-augment member refuses without fetching anything (the client keeps the reference's loader).  Every other dataset is
        PROBED -- the names only set how hard: a chain of known deterministic transforms (the reference's --resize / --imnorm: Resize, ToTensor,
        Normalize) fetches ONE sample twice, because randomness can also sit outside `transform` (a target_transform, a random choice inside
        __getitem__, a subclass that reuses a deterministic name); chains with members that cannot be judged by name (torchvision's Lambda --
        the reference pads its chains with identity Lambdas -- or user callables) fetch `probe_samples` samples `probe_repeats` times each.
        Image AND token tensors must come back identical (one pair of fetches alone, as in round 4,
        passes a RandomHorizontalFlip(0.5) every second time and then freezes ONE augmented view per image), with the torch / numpy / python
        RNG states saved before and restored after, so that a probe never shifts the shuffle or augmentation stream of the run."""
        base, _ = _resolve(dataset)
        if base is None or len(dataset) == 0:
            return False
        verdict = cls._transform_verdict(getattr(base, "transform", None))
        if verdict == "random":
            return False
        import random as _random
        st_t, st_n, st_p = torch.get_rng_state(), np.random.get_state(), _random.getstate()
        try:
            n = len(dataset)
            picks = sorted({0, n // 2, n - 1})[:max(1, probe_samples)] if verdict == "unknown" else [0]
            reps = probe_repeats if verdict == "unknown" else 2
            for i in picks:
                a = dataset[i]
                if not (torch.is_tensor(a[0]) and a[0].dtype == torch.float32 and a[0].dim() == 3 and torch.is_tensor(a[1])):
                    return False
                for _ in range(reps - 1):
                    b = dataset[i]
                    if not (torch.equal(a[0], b[0]) and torch.is_tensor(b[1]) and torch.equal(a[1], b[1])):
                        return False
            return True
        except Exception:                              # unreadable data: let the ordinary loader raise where the reference would
            return False
        finally:
            torch.set_rng_state(st_t); np.random.set_state(st_n); _random.setstate(st_p)

    def __len__(self):
        return len(self.dataset)

    def __getstate__(self):              # deep-copied / pickled clients: the lock does not travel
        d = dict(self.__dict__)
        d["_lock"] = None
        d["_lut_dev"] = {}
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._lock = threading.Lock()

    def __deepcopy__(self, memo):
        """A copied client shares the decoded store (read-only after build()) instead of duplicating hundreds of MB per copy."""
        new = DecodedCache.__new__(DecodedCache)
        new.__dict__.update(self.__dict__)
        new._lock = threading.Lock()
        new._lut_dev = {}
        memo[id(self)] = new
        return new

    def _base_index(self, i):
        return int(i) if self.index is None else int(self.index[i])

    # ------------------------------------------------------------------ build
    def build(self):
        if self.built:
            return self
        with self._lock:
            if not self.built:
                self._build()
        return self

    def _build(self):
        n = len(self.dataset)
        keys = [self.base.image_key(self._base_index(i)) for i in range(n)]
        first = {}
        for i, k in enumerate(keys):
            first.setdefault(k, i)
        uniq = list(first.keys())
        row_of = {k: r for r, k in enumerate(uniq)}
        self.row = np.asarray([row_of[k] for k in keys], dtype=np.int64)
        s0 = self.dataset[0]
        C, H, W = s0[0].shape
        self.tokens = torch.empty((n,) + tuple(s0[1].shape), dtype=s0[1].dtype)
        self.meta = [None] * n
        rep = set(first.values())                      # the sample that decodes each image
        # Host memory: an image is turned into uint8 codes (and verified bit for bit) as soon as it is decoded, inside the worker, so the build
        # holds ONE byte per pixel-channel plus a float image per worker thread -- not the whole client in fp32 with int64 temporaries beside it
        # (> 7 GB for a 6 000-image 224 x 224 client).  The table is chosen on the first image; an image that does not invert under
        # it keeps its floats, and if any did the store falls back to float32 (the verified images are regenerated from their codes: same bits).
        lut = None
        if self.store_kind in ("auto", "uint8") and C == 3:
            x0 = s0[0].numpy()
            for norm in self.norms:
                cand = _lut(norm)
                if self._codes(x0, cand.numpy()) is not None:
                    lut = cand
                    break
        lut_np = lut.numpy() if lut is not None else None
        u8 = np.empty((len(uniq), C, H, W), dtype=np.uint8) if lut is not None else None
        floats = {} if lut is not None else None       # row -> float image that did not invert
        imgs = torch.empty((len(uniq), C, H, W), dtype=torch.float32) if lut is None else None

        def job(i):
            if i in rep:
                item = self.dataset[i]
                r = int(self.row[i])
                if lut is None:
                    imgs[r].copy_(item[0])
                else:
                    code = self._codes(item[0].numpy(), lut_np)
                    if code is None:
                        floats[r] = item[0].clone()
                    else:
                        u8[r] = code
            else:                                      # tokens / ids only: no image decode (the dataset may offer it; else the full fetch)
                item = self.base.sample_without_image(self._base_index(i)) if hasattr(self.base, "sample_without_image") else self.dataset[i]
            self.tokens[i].copy_(torch.as_tensor(item[1]))
            self.meta[i] = tuple(item[2:])

        with ThreadPoolExecutor(self.workers) as pool:
            list(pool.map(job, range(n)))
        self.lut, self.u8 = None, None
        if lut is not None and not floats:
            self.lut, self.u8 = lut, torch.from_numpy(u8)
            self._u8_np, self._lut_np = u8, lut_np
        else:
            if self.store_kind == "uint8":
                raise ValueError("DecodedCache(store='uint8'): the decoded images are not uint8 values under a ToTensor / Normalize table")
            if imgs is None:                           # some images did not invert: floats for all (the inverted ones regenerate exactly)
                imgs = torch.empty((len(uniq), C, H, W), dtype=torch.float32)
                for r in range(len(uniq)):
                    if r in floats:
                        imgs[r].copy_(floats[r])
                    else:
                        imgs[r].copy_(torch.from_numpy(np.stack([np.take(lut_np[c], u8[r, c]) for c in range(3)])))
            self.f32 = imgs
        self.meta_cols = None
        if n and all(isinstance(v, (int, np.integer)) for m in self.meta for v in m):
            self.meta_cols = [torch.tensor([m[j] for m in self.meta], dtype=torch.int64) for j in range(len(self.meta[0]))]
        self.built = True
        logger.info("[DecodedCache] %d samples over %d images, store %s (%.1f MB)", n, len(uniq), "uint8" if self.lut is not None else "float32",
                    (self.u8.numel() if self.lut is not None else self.f32.numel() * 4) / 1e6)
        return self

    @staticmethod
    def _codes(x, lut_np):
        """uint8 codes [3, H, W] of ONE float image under the per-channel table, or None unless the table regenerates it bit for bit."""
        out = np.empty(x.shape, dtype=np.uint8)
        for c in range(3):
            t = lut_np[c]
            if not np.all(np.diff(t) > 0):
                return None
            code = np.searchsorted(t, x[c], side="left").clip(0, 255)
            if not np.array_equal(t[code], x[c]):
                return None
            out[c] = code.astype(np.uint8)
        return out

    # ------------------------------------------------------------------ fetch
    def _image(self, r):
        if self.lut is not None:
            return torch.from_numpy(np.stack([np.take(self._lut_np[c], self._u8_np[r, c]) for c in range(3)]))
        return self.f32[r].clone()

    def __getitem__(self, i):
        self.build()
        return (self._image(int(self.row[i])), self.tokens[i].clone()) + tuple(self.meta[i])

    # ---- device-side expansion (the MI355X path): the batch leaves the host as uint8 codes and becomes the transform's floats on the GPU
    def get_batch_raw(self, idxs, out=None):
        """Like get_batch, but the image field is the uint8 codes when the store is uint8 (a quarter of the bytes over PCIe; no host
        arithmetic): the consumer turns it into floats with ``expand_on_device`` after the copy (DevicePrefetcher does, on its stream)."""
        self.build()
        if self.lut is None:
            return self.get_batch(idxs, out)
        ii = np.asarray(idxs, dtype=np.int64)
        rows = torch.from_numpy(self.row[ii])
        if out is None:
            out = [torch.empty((len(ii),) + tuple(self.u8.shape[1:]), dtype=torch.uint8), torch.empty((len(ii),) + tuple(self.tokens.shape[1:]), dtype=self.tokens.dtype)]
            out += [torch.empty(len(ii), dtype=torch.int64) for _ in range(len(self.meta[0]) if self.meta else 0)]
        torch.index_select(self.u8, 0, rows, out=out[0])
        self._fill_rest(ii, out)
        return tuple(out)

    def expand_on_device(self, img, out=None):
        """uint8 codes [b, 3, H, W] on the device -> the dataset's float32 images (fc_image_u8_to_f32 on the current stream).  `out`: a float32
        tensor of the same shape to fill (a prefetcher with a fixed device ring passes one per slot); default a new tensor."""
        if img.dtype != torch.uint8:
            return img
        from .. import _lib
        dev = img.device
        lut = self._lut_dev.get(dev)
        if lut is None:
            lut = self._lut_dev[dev] = self.lut.to(dev).contiguous()
        img = img.contiguous()
        if out is None or out.shape != img.shape or out.dtype != torch.float32 or out.device != dev:
            out = torch.empty(img.shape, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().fc_image_u8_to_f32(_lib.ptr(img), _lib.ptr(lut), _lib.ptr(out), img.shape[0], img.shape[1], img.shape[2] * img.shape[3],
                                                 _lib.stream_ptr()))
        return out

    def _fill_rest(self, ii, out):
        torch.index_select(self.tokens, 0, torch.from_numpy(ii), out=out[1])
        if self.meta_cols is not None:
            for j, col in enumerate(self.meta_cols):
                torch.index_select(col, 0, torch.from_numpy(ii), out=out[2 + j])
        else:
            for j in range(len(out) - 2):
                out[2 + j].copy_(torch.as_tensor([self.meta[i][j] for i in ii]))

    def get_batch(self, idxs, out=None):
        """Stacked fields of the samples `idxs` (the dataset's tuple layout); with `out` (views of a pinned batch) gathered in place."""
        self.build()
        ii = np.asarray(idxs, dtype=np.int64)
        rows = self.row[ii]
        if out is None:
            img = torch.empty((len(ii),) + tuple((self.u8 if self.lut is not None else self.f32).shape[1:]), dtype=torch.float32)
            tok = torch.empty((len(ii),) + tuple(self.tokens.shape[1:]), dtype=self.tokens.dtype)
            nmeta = len(self.meta[0]) if self.meta else 0
            out = [img, tok] + [torch.empty(len(ii), dtype=torch.int64) for _ in range(nmeta)]
        if self.lut is not None:
            dst = out[0].numpy()
            for j, r in enumerate(rows):               # per (sample, channel) plane, contiguous on both sides: one pass uint8 -> the
                for c in range(3):                     # transform's float value, no temporaries (numpy releases the GIL inside take)
                    np.take(self._lut_np[c], self._u8_np[r, c], out=dst[j, c], mode="clip")
        else:
            torch.index_select(self.f32, 0, torch.from_numpy(rows), out=out[0])
        self._fill_rest(ii, out)
        return tuple(out)
