#!/usr/bin/env python
"""Large-tile grouped NT GEMM (csrc/fc_gemm_big.hip) at the client step's shapes: correctness against torch fp32 on the same bf16
inputs, and stand-alone timings beside the 128x128-tile kernel (fc_k_gemm).  Run under `rocprofv3 --kernel-trace` + tools/ktrace.py for
device-side durations.   usage: tools/gemm_big_bench.py [reps] [check]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ["FC_PROBES_LIB"] = "1"        # the large-tile kernel is an experiment of the tools build (python -m fedcola_amd.build --probes)
from fedcola_amd import _lib
L = _lib.lib(); P = _lib.ptr
_P, _I = C.c_void_p, C.c_int32
L.fc_k_gemm_big.restype = C.c_int
L.fc_k_gemm_big.argtypes = [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
check = len(sys.argv) > 2
sp = _lib.stream_ptr()
MI, MT = 12608, 2048
# name, N, K, epi (0 plain, 1 bias, 2 res, 4 gelu_sg, 5 mul)
shapes = [("fc1 fwd", 1536, 384, 4), ("qkv fwd", 1152, 384, 1), ("proj fwd", 384, 384, 2), ("fc2 fwd", 384, 1536, 2),
          ("fc2 dX", 1536, 384, 5), ("fc1 dX", 384, 1536, 0), ("qkv dX", 384, 1152, 0), ("proj dX", 384, 384, 0)]
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()


def ref(A, W, bias, inp, epi):
    y = A.float() @ W.float().t()
    if epi in (1, 2, 4): y = y + bias
    if epi == 2: y = y + inp.float()
    if epi == 5: y = y * inp.float()
    if epi == 4:
        cdf = 0.5 * (1 + torch.erf(y * 0.7071067811865476)); pdf = torch.exp(-0.5 * y * y) * 0.3989422804014327
        return y * cdf, cdf + y * pdf
    return y, None


def timeit(f):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, N, K, epi in shapes:
    A0, A1, W0, W1 = bf(MI, K), bf(MT, K), bf(N, K) * 0.1, bf(N, K) * 0.1
    C0, C1 = torch.empty(MI, N, device="cuda", dtype=torch.bfloat16), torch.empty(MT, N, device="cuda", dtype=torch.bfloat16)
    b0, b1 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    i0, i1 = bf(MI, N), bf(MT, N)
    o0, o1 = torch.empty_like(C0), torch.empty_like(C1)
    for bm in (256, 128, 2):
        def run(two=True):
            _lib.check(L.fc_k_gemm_big(P(A0), P(W0), P(C0), MI, P(A1) if two else None, P(W1), P(C1), MT if two else 0, N, K, epi, P(b0), P(b1),
                                       P(i0), P(i1), P(o0), P(o1), None, None, 1, bm, sp))
        if check:
            C0.zero_(); C1.zero_(); run(); torch.cuda.synchronize()
            for (A, W, b, i, Cm, o) in ((A0, W0, b0, i0, C0, o0), (A1, W1, b1, i1, C1, o1)):
                y, gp = ref(A, W, b, i, epi)
                err = float((Cm.float() - y).abs().max() / y.abs().max())
                assert err < 1.5e-2, (name, bm, err)
                if gp is not None:
                    e2 = float((o.float() - gp).abs().max() / gp.abs().max())
                    assert e2 < 1.5e-2, (name, bm, "gelu'", e2)
        us2, us1 = timeit(run), timeit(lambda: run(False))
        fl2, fl1 = 2.0 * (MI + MT) * N * K, 2.0 * MI * N * K
        print(f"{name:9s} N={N:4d} K={K:4d} epi={epi} BM={bm}: img+txt {us2:7.1f} us {fl2 / us2 / 1e6:7.1f} TF/s | img only {us1:7.1f} us {fl1 / us1 / 1e6:7.1f} TF/s"
              + ("  [checked]" if check else ""))
    # the 128x128-tile kernel on the image rows (bias epilogue; NT form)
    us = timeit(lambda: _lib.check(L.fc_k_gemm(1, 0, 1, 1, P(A0), P(W0), P(C0), MI, N, K, P(b0), 0, sp)))
    print(f"{name:9s} 128x128-tile kernel, img only, bias epilogue: {us:7.1f} us {2.0 * MI * N * K / us / 1e6:7.1f} TF/s")
