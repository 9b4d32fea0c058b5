"""Drive the product (fedcola_amd, HIP) on the golden cases -- shared by the GPU parity tests and smoke()."""
import ctypes as C

import torch

import golden_util as G


def build_product(case_mk, precision="fp32", weights=None):
    from fedcola_amd.mome import ModalityAgnosticTransformer as M
    mk = dict(case_mk)
    m = M(precision=precision, init=False, **mk)
    if weights is not None:
        m.load_state_dict(weights, strict=True)
    return m.cuda()


def product_step(model, kind, img, ids, y, lr, droppath=None, wd=0.0, step=1, state=None, prepare=True):
    """One fc_client_step; returns (loss, grads_by_key, state)."""
    from fedcola_amd import _lib
    dev = model.flat.device
    n = model.flat.numel()
    if state is None:
        state = dict(grads=torch.zeros(n, device=dev), m=torch.zeros(n, device=dev), v=torch.zeros(n, device=dev),
                     loss=torch.zeros(2, device=dev))
    B = (img if kind != "txt" else ids).shape[0]
    n_txt = ids.shape[1] if kind != "img" else 0
    if prepare:           # False: keep the bf16 compute weights the previous step's optimizer wrote
        model.prepare_weights(force=True)
    ws = model.workspace(B, n_txt)
    imgd = img.cuda().contiguous() if kind != "txt" else None
    idsd = ids.cuda().contiguous() if kind != "img" else None
    yd = y.cuda().contiguous() if kind != "img+txt" else None
    P = _lib.ptr
    _lib.check(_lib.lib().fc_client_step(model._handle.h, P(model.flat), P(state["grads"]), P(state["m"]), P(state["v"]),
                                         P(model._wc_or_flat()), P(imgd), P(idsd), P(yd), B, n_txt, P(droppath), lr, 0.9, 0.999, 1e-8,
                                         wd, step, P(state["loss"]), P(ws), ws.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    grads = {}
    for k, s in model.segments.items():
        grads[k] = state["grads"][s["offset"]: s["offset"] + s["numel"]].view(s["shape"]).cpu()
    return float(state["loss"][1]), grads, state
