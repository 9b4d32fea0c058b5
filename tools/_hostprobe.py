import time, torch, os
from concurrent.futures import ThreadPoolExecutor
n = 512
img = torch.randn(n, 3, 224, 224); ids = torch.randint(0, 7732, (n, 32))
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
def T(f, reps=5):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3
idx = torch.randperm(n)[:64]
print("alloc pinned 38.5MB: %.2f ms" % T(lambda: torch.empty(64, 3, 224, 224, pin_memory=True)))
print("alloc pageable 38.5MB: %.2f ms" % T(lambda: torch.empty(64, 3, 224, 224)))
buf = torch.empty(64, 3, 224, 224, pin_memory=True)
def loop1():
    for j, i in enumerate(idx.tolist()): buf[j].copy_(img[i])
print("per-sample copy loop, 1 thread: %.2f ms" % T(loop1))
for w in (4, 8, 16, 32):
    pool = ThreadPoolExecutor(w)
    def loopw():
        list(pool.map(lambda ji: buf[ji[0]].copy_(img[ji[1]]), enumerate(idx.tolist())))
    print("per-sample copy, %d threads: %.2f ms" % (w, T(loopw)))
    def chunked():
        il = idx.tolist(); k = (64 + w - 1) // w
        def job(c):
            for j in range(c * k, min(64, (c + 1) * k)): buf[j].copy_(img[il[j]])
        list(pool.map(job, range(w)))
    print("chunked copy, %d threads: %.2f ms" % (w, T(chunked)))
print("gather img[idx]: %.2f ms" % T(lambda: img[idx]))
print("index_select out=pinned: %.2f ms" % T(lambda: torch.index_select(img, 0, idx, out=buf)))
for t in (1, 8, 32):
    torch.set_num_threads(t)
    print("index_select out=pinned, torch threads %d: %.2f ms" % (t, T(lambda: torch.index_select(img, 0, idx, out=buf))))
    print("stack (default_collate core), torch threads %d: %.2f ms" % (t, T(lambda: torch.stack([img[i] for i in idx.tolist()]))))
d = torch.empty(64, 3, 224, 224, device="cuda")
torch.cuda.synchronize()
def h2d():
    d.copy_(buf, non_blocking=True); torch.cuda.synchronize()
print("H2D 38.5MB pinned: %.2f ms" % T(h2d))
