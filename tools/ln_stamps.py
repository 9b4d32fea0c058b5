#!/usr/bin/env python
"""In-kernel timeline of the LayerNorm backward (tools build only).  usage: FC_PROBES_LIB=1 python tools/ln_stamps.py [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FC_PROBES_LIB", "1")
import numpy as np, torch
from fedcola_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M, D = B * 197, 384
raw = C.CDLL(_lib.LIB_PATH); L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr()
bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()
x, dy, res, dx = bf(M, D), bf(M, D), bf(M, D), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
g = torch.randn(D, device="cuda"); mean, rstd = torch.randn(M, device="cuda"), torch.rand(M, device="cuda")
dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
part = torch.empty(int(L.fc_k_layernorm_partial_floats(M, D)), device="cuda")
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for it in range(4):
    big.fill_(it)                      # evict L2 / Infinity Cache: the operands come from HBM like (most of) them do in the step
    torch.cuda.synchronize()
    _lib.check(L.fc_k_layernorm_bwd_partial(1, P(dy), P(x), P(mean), P(rstd), P(g), P(res), P(dx), P(dg), P(db), M, D, P(part), sp))
    torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.int64)
raw.fc_dbg_ln_read_stamps(buf.ctypes.data_as(C.c_void_p))
st = buf.reshape(1024, 8)
nb = min((M + 15) // 16, 1024)
st = st[:nb]
t0 = st[:, 0].min()
names = ["start->loads issued+landed(1)", "compute(2)", "shuffles(3)", "dx stores issued(4)", "LDS+sync(5)", "partial store(6)"]
d = np.diff(st[:, :7], axis=1) * 10.0   # ns
print(f"M={M}: {nb} blocks; first start .. last end: {(st[:, 6].max() - t0) * 10 / 1e3:.2f} us; block start spread {(st[:, 0].max() - t0) * 10 / 1e3:.2f} us")
for i, n in enumerate(names):
    print(f"  {n:34s} mean {d[:, i].mean():8.0f} ns   p10 {np.percentile(d[:, i], 10):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
print(f"  block lifetime mean {(st[:, 6] - st[:, 0]).mean() * 10:.0f} ns")
