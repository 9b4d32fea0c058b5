#!/usr/bin/env python
"""GPU time of the client step's phases (tools build: FC_PROBES_LIB=1 FC_STEP_PHASES=1), hipEvents on the caller's stream -- no
profiler in the way.  usage: tools/step_phases.py [steps]"""
import ctypes as C, os, sys
os.environ.setdefault("FC_PROBES_LIB", "1"); os.environ.setdefault("FC_STEP_PHASES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import Args, make_batch
from fedcola_amd import _lib
from fedcola_amd.mome import create_model
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
raw = C.CDLL(_lib.LIB_PATH)
a = Args(); dev = torch.device("cuda")
model = create_model("mome_small_patch16", False, args=a, num_classes=[None, None], modalities=["img", "txt"], tasks=["rtv", "rtv"]).to(dev)
model.train()
img, ids = make_batch(64, 32, 7732, 0, dev)
n = model.flat.numel()
grads, m1, m2 = (torch.zeros(n, device=dev) for _ in range(3)); lossbuf = torch.zeros(2, device=dev)
model.prepare_weights(force=True); ws = model.workspace(64, 32)
L, P = _lib.lib(), _lib.ptr
out = (C.c_float * 5)()
W = 15
for k in range(1, W + steps + 1):
    _lib.check(L.fc_client_step(model._handle.h, P(model.flat), P(grads), P(m1), P(m2), P(model._wc_or_flat()), P(img), P(ids), None, 64, 32, None,
                                1e-4, 0.9, 0.999, 1e-8, 0.0, k, P(lossbuf), P(ws), ws.numel(), _lib.stream_ptr()))
torch.cuda.synchronize()
acc = np.zeros(5)
for k in range(W + 1, W + steps + 1):        # steady state: no synchronisation between the steps (steps <= 48: event ring of 64)
    raw.fc_dbg_step_phases(k, out)
    acc += np.array(list(out))
acc /= steps
ev = np.zeros(7); o7 = (C.c_float * 7)()
for k in range(W + 1, W + steps + 1):
    raw.fc_dbg_stream_events(k, o7)
    ev += np.array(list(o7))
ev /= steps
print("ms from step start to the end of each stream's part: forward text %.3f  chain1 %.3f  chain0 %.3f | backward text %.3f  chain1 %.3f  chain0 %.3f | last dW chunk %.3f"
      % tuple(ev))
print("steady-state phases (ms): forward %.3f  loss %.3f  backward %.3f  optimizer+late dW %.3f  idle gap before the step %.3f  -> %.3f per step"
      % (*acc, acc.sum()))
