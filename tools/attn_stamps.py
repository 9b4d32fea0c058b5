#!/usr/bin/env python
"""In-kernel phase stamps of the attention backward kernel (tools build): python tools/attn_stamps.py [rows] [cold]
Prints, for the workgroup of (batch 0, head 0) of each body (dQ / dK-dV), per wave the cycles (s_memtime) of: staging + barrier,
each 16-row block's fragment loads, its pair loop and its stores.  cold = 1: a >= 1-GB ring of operands ahead of the stamped launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FC_PROBES_LIB", "1"); os.environ.setdefault("FC_ATTN_STAMPS", "1")
import torch
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr; sp = _lib.stream_ptr(); ck = _lib.check
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12608
cold = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, H, D = 197, 6, 384
B = M // N
bf = torch.bfloat16
nset = 24 if cold else 1
QKV = [torch.randn(B * N, 3 * D, device="cuda").to(bf) for _ in range(nset)]
O = [torch.randn(B * N, D, device="cuda").to(bf) for _ in range(nset)]; DO = [torch.randn(B * N, D, device="cuda").to(bf) for _ in range(nset)]
DQKV = [torch.empty(B * N, 3 * D, device="cuda", dtype=bf) for _ in range(nset)]
lse = torch.zeros(B * H * N, device="cuda"); delta = torch.zeros(B * H * N, device="cuda")
for i in range(nset + 3):
    k = i % nset
    ck(L.fc_k_attention_bwd(1, 1, P(QKV[k]), P(O[k]), P(DO[k]), P(lse), P(delta), P(DQKV[k]), B, N, H, 64, 0.125, sp))
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_longlong * 1024)()
assert raw.fc_dbg_attn_stamps(buf) == 0
print(f"# rows {M} ({B} x {N}), {'cold ring' if cold else 'warm'}; cycles (s_memtime, 100 MHz ticks x ? -- relative) of the (batch 0, head 0) workgroups")
for body, name in ((0, "dQ body (K, V in LDS)"), (1, "dK/dV body (Q, dO in LDS)")):
    st = [[buf[body * 512 + w * 64 + k] for k in range(64)] for w in range(8)]
    t0 = min(st[w][0] for w in range(8) if st[w][0])
    print(name)
    for w in range(8):
        s = st[w]
        line = f"  wave {w}: start {s[0]-t0:6d} | staging + barrier {s[1]-s[0]:6d}"
        for r in range(2):
            a, b, c = s[2 + 4 * r], s[3 + 4 * r], s[4 + 4 * r]
            if a: line += f" | block {r}: fragments {b-a:6d} loop {c-b:6d}"
        line += f" | end at {s[20]-t0:6d}"
        print(line)
