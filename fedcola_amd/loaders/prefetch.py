"""Device prefetcher for the fused client step.

The reference moves every batch with a blocking ``.to(device)`` inside the training loop (src/client/fedavgclient.py:82-93) from
a single-process ``DataLoader``.  Once a step is a few milliseconds, a B=64 batch of 224x224 fp32 images (38.5 MB, ~0.6 ms over
PCIe Gen5) must not sit on the step's critical path: batches are staged through reusable pinned buffers and copied on a side
HIP stream ``depth`` batches ahead; the consumer's stream waits on the copy's event only.

Which side stream matters on this GPU: a process drives four hardware queues well, and the client step already uses four
(caller, text tower, second image chain, weight gradients); a fifth stream for copies made the ViT-S step 7.4 ms instead of
5.6 ms.  Pass ``stream=model.side_stream()`` -- the text tower's stream, idle for most of the step -- and the copy of the next
batch runs behind the text tower's backward, under the image tower's."""
from __future__ import annotations

from collections import deque

import torch


class DevicePrefetcher:
    def __init__(self, loader, device="cuda", depth: int = 2, stream=None, reuse_device: bool = False, device_ring=None):
        """reuse_device: the batches are copied into a fixed ring of depth + 1 device buffers per position instead of freshly allocated
        tensors, so that the consumer sees the SAME addresses every depth + 1 batches (fc_client_step replays a captured HIP graph per
        set of addresses).  A yielded batch is then only valid until depth + 1 further batches have been taken: for consumers that use a
        batch before they ask for the next one (the training loops), not for `list(prefetcher)`.  device_ring: a dict that outlives this
        object (a client keeps one per loader) so that the addresses also survive from one epoch / round to the next."""
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.stream = stream                   # torch.cuda.Stream / ExternalStream for the copies (default: a new side stream)
        self._pinned = {}                      # (slot, position) -> reusable pinned staging tensor
        self._it = None
        self.reuse_device = bool(reuse_device) or device_ring is not None
        self._dev = device_ring if device_ring is not None else {}      # (slot, position) -> reusable device tensor

    def start(self):
        """Create the underlying loader's iterator NOW (a PinnedBatchLoader starts assembling its first batches at that point, and any
        sampler draws its RNG numbers here) and keep it for the coming __iter__: lets the caller overlap its own set-up with the first
        batch.  Returns self."""
        self._it = iter(self.loader)
        return self

    def __len__(self):
        return len(self.loader)

    def _to_device(self, slot, j, src):
        if not self.reuse_device:
            return src.to(self.device, non_blocking=True)
        dst = self._dev.get((slot, j))
        if dst is None or dst.shape != src.shape or dst.dtype != src.dtype or dst.device != self.device:
            dst = torch.empty(src.shape, dtype=src.dtype, device=self.device)
            self._dev[(slot, j)] = dst
        dst.copy_(src, non_blocking=True)
        return dst

    def _stage(self, slot, batch, stream):
        out = []
        with torch.cuda.stream(stream):
            for j, t in enumerate(batch):
                if not torch.is_tensor(t):
                    out.append(t)
                    continue
                if t.is_cuda:
                    out.append(t)
                    continue
                if t.is_pinned():                                     # DataLoader(pin_memory=True): no staging copy needed
                    out.append(self._to_device(slot, j, t))
                    continue
                key = (slot, j)
                buf = self._pinned.get(key)
                if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                    buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    self._pinned[key] = buf
                buf.copy_(t)                                          # pageable -> pinned (the only host-side copy)
                out.append(self._to_device(slot, j, buf))             # pinned -> HBM on the copy stream
            finish = getattr(self.loader, "device_finish", None)      # a raw loader's uint8 image codes -> floats, behind the copy
            if finish is not None:
                if self.reuse_device and out and torch.is_tensor(out[0]) and out[0].dtype == torch.uint8:
                    key = (slot, "expanded")
                    dst = self._dev.get(key)
                    if dst is None or dst.shape != out[0].shape or dst.device != out[0].device:
                        dst = self._dev[key] = torch.empty(out[0].shape, dtype=torch.float32, device=out[0].device)
                    out = list(finish(out, out=dst))
                else:
                    out = list(finish(out))
            ev = torch.cuda.Event()
            ev.record(stream)
        return out, ev

    def __iter__(self):
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher stages batches into GPU memory; device must be a cuda device")
        stream = self.stream if self.stream is not None else torch.cuda.Stream(device=self.device)
        pending = deque()
        nslots = self.depth + 1                # a slot's pinned buffers are reused only after its batch has been consumed
        it, self._it = (self._it if self._it is not None else iter(self.loader)), None
        src = it
        slot = 0
        done_events = {}                       # slot -> event after which its pinned buffers may be overwritten
        consumed = {}                          # slot -> event on the CONSUMER's stream after which its device buffers may be overwritten
        first = True
        try:
            while True:
                # the first batch is handed over as soon as IT is staged (waiting for `depth` batches up front costs the consumer a
                # batch-assembly time per epoch: 2-3 ms of a 100-ms client round); from then on `depth` batches are kept in flight
                while it is not None and len(pending) < (1 if first else self.depth):
                    try:
                        batch = next(it)
                    except StopIteration:
                        it = None
                        break
                    if slot in done_events:
                        done_events.pop(slot).synchronize()
                    if slot in consumed:          # fixed device ring: the copy stream waits for the step that read this slot's buffers
                        stream.wait_event(consumed.pop(slot))
                    pending.append((slot,) + self._stage(slot, list(batch), stream))
                    slot = (slot + 1) % nslots
                first = False
                if not pending:
                    return
                s, tensors, ev = pending.popleft()
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in tensors:
                    if torch.is_tensor(t):
                        t.record_stream(torch.cuda.current_stream(self.device))
                done_events[s] = ev
                yield tensors
                if self.reuse_device:             # the consumer has enqueued its work on this batch by the time it asks for the next
                    cev = torch.cuda.Event()
                    cev.record(torch.cuda.current_stream(self.device))
                    consumed[s] = cev
        finally:
            pending.clear()
            if hasattr(src, "close"):          # an abandoned epoch (debug break): stop the loader's producer thread
                src.close()
