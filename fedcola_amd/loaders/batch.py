"""Host-side batch assembly for the client loops.

The reference iterates a single-process ``DataLoader(dataset, batch_size, shuffle)`` (src/client/fedavgclient.py:44-53): 64 samples
are fetched one by one and ``default_collate`` stacks them into a new pageable tensor -- ~100 ms for a B = 64 batch of 224x224
fp32 images on the GPU box's host, 18x the device step.  ``PinnedBatchLoader`` keeps the reference's sampling exactly (torch's
``RandomSampler`` / ``SequentialSampler`` + ``BatchSampler``: the same index order under the same RNG state) but has a few worker
threads (each limited to ONE intra-op thread) write contiguous chunks of samples straight into a pinned batch buffer (tensor copies
release the GIL); the buffers come from torch's
caching pinned allocator, which does not recycle a block while an asynchronous H2D copy from it is in flight, so the batches can
be handed to ``DevicePrefetcher`` as they are.  A dataset may offer ``get_batch(indices) -> tuple of stacked fields`` (same fields as
``__getitem__``; with an ``out=`` parameter the fields are gathered straight into the pinned batch); then the per-sample Python
overhead (the remaining ~20 ms per batch) disappears as well.

Reproducibility: the index order is the DataLoader's.  Samples are fetched by worker threads, so a dataset whose ``__getitem__`` draws
from the global torch RNG (random-crop / flip transforms) sees those draws in a thread-dependent order: ``FedavgClient`` therefore
uses this loader by default only for datasets that offer ``get_batch`` (in-memory / pre-decoded, deterministic per index) and keeps
the reference's DataLoader otherwise (``args.fast_loader = True`` opts in explicitly)."""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler


def _worker_init():
    # A tensor copy inside a pool thread would otherwise fan out over torch's intra-op pool (128 threads on the MI355X host): dozens
    # of pool threads each forking 128-way for a 600-KB copy is what made threaded assembly slower than a plain loop (measured
    # 3-32 ms per batch, against 4.8 ms for ONE thread and 1.5 ms for 4 single-threaded workers on contiguous chunks).
    torch.set_num_threads(1)


class PinnedBatchLoader:
    def __init__(self, dataset, batch_size: int, shuffle: bool = False, drop_last: bool = False, workers: int = 4, pin: bool = True, ahead: int = 2,
                 raw: bool = False):
        """raw: batches come from ``dataset.get_batch_raw`` (a DecodedCache's uint8 image codes: a quarter of the bytes, no host
        arithmetic) and must pass through ``loader.device_finish`` after their copy to the device -- DevicePrefetcher does that on its
        copy stream.  Only for consumers that do (FedavgClient's training loop)."""
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.workers, self.pin, self.ahead = max(1, int(workers)), pin and torch.cuda.is_available(), max(0, int(ahead))
        self._spec = None
        self._pool = None
        self.raw = bool(raw) and hasattr(dataset, "get_batch_raw") and hasattr(dataset, "expand_on_device")
        gb = getattr(dataset, "get_batch_raw" if self.raw else "get_batch", None)
        self._fetch = gb
        self._into = False
        if gb is not None:
            import inspect
            try:
                self._into = "out" in inspect.signature(gb).parameters
            except (TypeError, ValueError):
                pass

    def _get_pool(self):
        # one pool for the loader's lifetime: starting 8 threads and running torch.set_num_threads(1) in each costs 2-3 ms, which
        # an epoch of 20 steps (100 ms) pays again at every iteration otherwise
        if self._pool is None:
            self._pool = ThreadPoolExecutor(self.workers, initializer=_worker_init)
            list(self._pool.map(lambda _: None, range(self.workers)))      # start the threads now
        return self._pool

    def __getstate__(self):              # copies / pickles of a loader (deep-copied clients) start without the pool
        d = dict(self.__dict__)
        d["_pool"] = None
        d["_fetch"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._fetch = getattr(self.dataset, "get_batch_raw" if self.raw else "get_batch", None)

    def device_finish(self, tensors, out=None):
        """After the copy to the device (on the copy's stream): raw image codes -> the dataset's float images.  Identity otherwise.
        out: optional float32 tensor to expand into (DevicePrefetcher's fixed device ring)."""
        if self.raw and tensors and torch.is_tensor(tensors[0]) and tensors[0].dtype == torch.uint8:
            tensors = list(tensors)
            tensors[0] = self.dataset.expand_on_device(tensors[0], out=out)
        return tensors

    def __del__(self):
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _fill_chunk(self, bufs, j0, idxs):
        for j, i in enumerate(idxs, j0):
            item = self.dataset[i]
            for f, v in enumerate(item):
                bufs[f][j].copy_(torch.as_tensor(v))

    def __iter__(self):
        return _BatchIter(self)


class _BatchIter:
    """One epoch of a PinnedBatchLoader.  An object, not a generator: the RNG draws and the producer thread start when iter() is CALLED,
    so a caller may create the iterator early and let the first batch be assembled under its own set-up work (FedavgClient.update()
    starts it before it allocates the optimizer state: 1 ms of a 100-ms round)."""

    def __init__(self, ld: PinnedBatchLoader):
        self.ld = ld
        # DataLoader.__iter__ draws its base seed from the default RNG before the sampler draws the permutation seed: consume the
        # same number so that a shuffled epoch visits the samples in exactly the DataLoader's order under the same RNG state
        torch.empty((), dtype=torch.int64).random_()
        sampler = RandomSampler(ld.dataset) if ld.shuffle else SequentialSampler(ld.dataset)
        self.batches = list(BatchSampler(sampler, ld.batch_size, ld.drop_last))      # every RNG draw happens here, in the caller's thread
        self.pool = ld._get_pool()
        self.pos = 0
        self.q = None
        self.stop = threading.Event()
        self.thread = None
        if ld.ahead > 0:
            # a producer thread assembles `ahead` batches in advance, under the consumer's device work
            self.q = queue.Queue(maxsize=ld.ahead)
            self.thread = threading.Thread(target=self._produce, daemon=True)
            self.thread.start()

    def _assemble(self, idxs):
        ld, pool, W = self.ld, self.pool, self.ld.workers
        n = len(idxs)
        k = (n + W - 1) // W
        chunks = [(c * k, idxs[c * k: (c + 1) * k]) for c in range(W) if c * k < n]
        if hasattr(ld.dataset, "get_batch"):          # vectorised fetch (in-memory / pre-decoded datasets): one gather per field and chunk
            if ld._spec is None:                       # field shapes / dtypes: asked once per loader, not once per batch
                ld._spec = [(tuple(t.shape[1:]), t.dtype) for t in map(torch.as_tensor, ld._fetch(idxs[:1]))]
            bufs = [torch.empty((n,) + sh, dtype=dt, pin_memory=ld.pin) for sh, dt in ld._spec]

            def job(ch):
                j0, ii = ch
                views = [buf[j0: j0 + len(ii)] for buf in bufs]
                if ld._into:                           # get_batch(indices, out=views): gathered straight into the pinned batch (one copy)
                    ld._fetch(ii, out=views)
                else:
                    for v, t in zip(views, ld._fetch(ii)):
                        v.copy_(torch.as_tensor(t))
            list(pool.map(job, chunks))
            return tuple(bufs)
        first = [torch.as_tensor(v) for v in ld.dataset[idxs[0]]]
        bufs = [torch.empty((n,) + tuple(t.shape), dtype=t.dtype, pin_memory=ld.pin) for t in first]
        for f, v in enumerate(first):                    # the sample fetched for the shapes is used, not fetched again (one decode, one
            bufs[f][0].copy_(v)                          # set of RNG draws per sample, like the DataLoader)
        chunks = [(j0, ii) if j0 else (1, ii[1:]) for j0, ii in chunks]
        list(pool.map(lambda ch: ld._fill_chunk(bufs, ch[0], ch[1]), chunks))
        return tuple(bufs)

    def _put(self, item):
        while not self.stop.is_set():   # never blocks for good: an abandoned epoch (close()) lets the thread end
            try:
                self.q.put(item, timeout=0.05)
                return
            except queue.Full:
                pass

    def _produce(self):
        try:
            for idxs in self.batches:
                if self.stop.is_set():
                    return
                self._put(self._assemble(idxs))
            self._put(None)
        except BaseException as e:      # surfaces in the consumer
            self._put(e)

    def __iter__(self):
        return self

    def __next__(self):
        if self.q is None:
            if self.pos >= len(self.batches):
                raise StopIteration
            self.pos += 1
            return self._assemble(self.batches[self.pos - 1])
        while True:                     # after close() the producer ends without a sentinel: never block on an abandoned epoch
            try:
                item = self.q.get(timeout=0.05)
                break
            except queue.Empty:
                if self.stop.is_set() and (self.thread is None or not self.thread.is_alive()):
                    raise StopIteration
        if item is None:
            self.q.put(None)            # stays exhausted
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        self.stop.set()
        t, self.thread = self.thread, None
        while t is not None and t.is_alive():   # unblock a producer waiting on a full queue
            try:
                self.q.get_nowait()
            except queue.Empty:
                pass
            t.join(timeout=0.05)

    def __del__(self):
        try:
            self.stop.set()
        except Exception:
            pass
