#!/usr/bin/env python
"""In-kernel timeline of the weight-stationary GEMM (tools build only: FC_PROBES_LIB=1 FC_WS_STAMPS=1).
usage: FC_PROBES_LIB=1 FC_WS_STAMPS=1 python tools/ws_stamps.py kind M N K"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FC_PROBES_LIB", "1"); os.environ.setdefault("FC_WS_STAMPS", "1")
import numpy as np, torch
from fedcola_amd import _lib
kind, M, N, K = (int(x) for x in sys.argv[1:5])
raw = C.CDLL(_lib.LIB_PATH)
L = _lib.lib(); P = _lib.ptr
A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(*((N, K) if kind == 0 else (K, N)), device="cuda").bfloat16()
Cm = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); bias = torch.randn(N, device="cuda")
sp = _lib.stream_ptr()
for _ in range(20):
    _lib.check(L.fc_k_gemm(1, kind, 1, 1, P(A), P(B), P(Cm), M, N, K, P(bias), 0, sp))
torch.cuda.synchronize()
G = 256
buf = np.zeros(G * 256, dtype=np.int64)
raw.fc_dbg_ws_read_stamps(buf.ctypes.data_as(C.c_void_p), G * 256)
st = buf.reshape(G, 256)
n = st[:, 255]
print("stamps per WG: min %d max %d" % (n.min(), n.max()))
for wg in (0, 1, 8, 100, 255):
    s = st[wg]; k = int(s[255])
    if k < 4: continue
    real = (s[1] - s[0]) * 10.0      # ns (100 MHz)
    cyc = s[k - 1] - s[2]
    print(f"WG {wg}: {k} stamps, wall {real/1e3:.2f} us, {cyc} cycles -> clock {cyc/real:.2f} GHz")
    d = np.diff(s[2:k])
    print("   deltas:", " ".join(str(int(v)) for v in d[:70]))
# per-step phase averages over all WGs with the modal stamp count (5 stamps per step: top, wait, barrier, [issue+compute+half], end)
