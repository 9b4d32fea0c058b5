#!/usr/bin/env python
"""Dump per-workgroup s_memtime stamps of one MFMA GEMM launch (development aid)."""
import ctypes as C, os, sys
os.environ["FC_GEMM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
kind, M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
shpA = (M, K) if kind != 2 else (K, M); shpB = (N, K) if kind == 0 else (K, N)
A = torch.randn(*shpA, device="cuda").bfloat16(); B = torch.randn(*shpB, device="cuda").bfloat16()
Cm = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); bias = torch.randn(N, device="cuda")
for _ in range(3):
    _lib.check(L.fc_k_gemm(1, kind, 1, 1, P(A), P(B), P(Cm), M, N, K, P(bias), 0, _lib.stream_ptr()))
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
n = 4096 * 32
buf = (C.c_longlong * n)()
raw.fc_dbg_read_stamps(buf, n)
st = np.frombuffer(buf, dtype=np.int64).reshape(4096, 32)
used = st[:, 0] != 0
st = st[used]
print("workgroups", st.shape[0])
t0 = st[:, 0].min()
nz = (st != 0).sum(1)
print("stamps per wg", np.unique(nz))
rel = np.where(st != 0, st - t0, -1)
for w in [0, 1, st.shape[0] // 2, st.shape[0] - 1]:
    r = rel[w][rel[w] >= 0]
    print("wg", w, "start", r[0], "deltas", np.diff(r).tolist())
d = np.diff(np.where(st != 0, st, 0), axis=1)
k = nz.min()
print("median deltas over wgs:", [int(np.median(st[:, i + 1] - st[:, i])) for i in range(k - 1)])
print("kernel span (cycles):", int(st.max() - t0))
