#!/usr/bin/env python
"""In-kernel phase stamps of the fused MLP kernel (tools build): FC_PROBES_LIB=1 FC_MLP_STAMPS=1 python tools/mlp_stamps.py [rows] [bwd]
Prints, for workgroup 0, per role (wave 0 = first product + activation, wave 4 = second product) the cycles of each phase per chunk."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FC_PROBES_LIB", "1"); os.environ.setdefault("FC_MLP_STAMPS", "1")
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr(); ck = _lib.check
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bwd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
D, Hd = 384, 1536
bf = torch.bfloat16
X = torch.randn(M, D, device="cuda").to(bf); W1 = (torch.randn(Hd, D, device="cuda") * D ** -0.5).to(bf); W2 = (torch.randn(D, Hd, device="cuda") * Hd ** -0.5).to(bf)
b1 = torch.randn(Hd, device="cuda") * 0.1; b2 = torch.randn(D, device="cuda") * 0.1; res = torch.randn(M, D, device="cuda").to(bf)
act = torch.empty(M, Hd, device="cuda", dtype=bf); gs = torch.randn(M, Hd, device="cuda").to(bf); out = torch.empty(M, D, device="cuda", dtype=bf)
pf = torch.empty(2 * D * Hd, device="cuda", dtype=bf); pb = torch.empty_like(pf)
ck(L.fc_k_mlp_pack(P(W1), P(W2), P(pf), P(pb), D, Hd, sp))
raw = C.CDLL(_lib.LIB_PATH)
for rep in range(3):
    if bwd: ck(L.fc_k_mlp_fused(1, P(X), P(pb), None, None, P(act), P(gs), None, None, 1, P(out), M, D, Hd, sp))
    else: ck(L.fc_k_mlp_fused(0, P(X), P(pf), P(b1), P(b2), P(act), P(gs), P(res), None, 32, P(out), M, D, Hd, sp))
    torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
assert raw.fc_dbg_mlp_stamps(buf) == 0
st = [[buf[w * 64 + k] for k in range(64)] for w in range(8)]
t0 = min(st[w][0] for w in range(8))
print(f"# rows {M}, {'backward' if bwd else 'forward'}; cycles (s_memtime) relative to the first wave's stamp 0; workgroup 0")
for w in (0, 4):
    s = st[w]
    print(f"wave {w} ({'first product + activation' if w < 4 else 'second product'}): X landed at {s[0]-t0}, barrier passed {s[1]-t0}, output phase {s[60]-t0} .. {s[61]-t0}")
    for c in range(12):
        a, b, d, e = s[2 + 4 * c], s[3 + 4 * c], s[4 + 4 * c], s[5 + 4 * c]
        if w < 4: print(f"   chunk {c:2d}: start {a-t0:7d} | 6 pieces {b-a:6d} | phase barrier + activation {d-b:6d} | barrier wait {e-d:6d}")
        else: print(f"   chunk {c:2d}: start {a-t0:7d} | image stores + first reads {d-a:6d} | phase barrier + 6 pieces {b-d:6d} | barrier wait {e-b:6d}")
