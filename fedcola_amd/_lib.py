"""ctypes binding of libfedcola_hip.so (the C ABI declared in include/fedcola_hip.h).

The product path has no CPU or PyTorch fallback: if the HIP library is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfedcola_hip.so")
if os.environ.get("FC_LIB_PATH"):         # tools only: A/B runs of two builds on one GPU box (tools/ab_bench.sh)
    LIB_PATH = os.environ["FC_LIB_PATH"]
elif os.environ.get("FC_PROBES_LIB"):      # tools only: the -DFC_PROBES build with the measurement aids (python -m fedcola_amd.build --probes)
    LIB_PATH = os.path.join(_HERE, "libfedcola_hip_probes.so")

FC_PREC_FP32, FC_PREC_BF16 = 0, 1
FC_TASK_NONE, FC_TASK_CLS, FC_TASK_RTV = 0, 1, 2
FC_OPT_MLP_FUSED, FC_OPT_STEP_GRAPH, FC_OPT_GEMM_FORM = 1, 2, 3


class FcModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "has_img", "has_txt", "img_size", "patch", "in_chans", "dim", "depth", "heads", "mlp_hidden", "vocab",
        "max_text_len", "task_img", "task_txt", "num_classes_img", "num_classes_txt", "with_aux", "aux_trained",
        "aux_attn_only", "aux_mlp_only", "precision", "colearn_attn")]


class FcSegment(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("offset", C.c_int64), ("numel", C.c_int64), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4), ("trainable", C.c_int32)]


class FedcolaHipError(RuntimeError):
    pass


_lib = None

_P = C.c_void_p
_I = C.c_int32
_F = C.c_float
_L = C.c_int64
_Z = C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/fedcola_hip.h declares (tests check this).
SIGNATURES = {
    "fc_last_error": (C.c_char_p, []),
    "fc_abi_version": (C.c_int, []),
    "fc_model_create": (C.c_int, [C.POINTER(FcModelCfg), C.POINTER(_P)]),
    "fc_model_destroy": (None, [_P]),
    "fc_model_num_params": (_L, [_P]),
    "fc_model_num_segments": (_I, [_P]),
    "fc_model_segment": (C.c_int, [_P, _I, C.POINTER(FcSegment)]),
    "fc_model_set_trainable": (C.c_int, [_P, _I, _I]),
    "fc_workspace_bytes": (_Z, [_P, _I, _I]),
    "fc_compute_weights_bytes": (_Z, [_P]),
    "fc_prepare_weights": (C.c_int, [_P, _P, _P, _P]),
    "fc_forward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _Z, _P, _P, _P]),
    "fc_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "fc_contrastive_loss_fwd_bwd": (C.c_int, [_P, _P, _I, _I, _F, _P, _Z, _P, _P, _P, _P]),
    "fc_contrastive_scratch_floats": (_Z, [_I]),
    "fc_ce_loss_fwd_bwd": (C.c_int, [_P, _P, _I, _I, _P, _P, _P]),
    "fc_adamw_step": (C.c_int, [_P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _P]),
    "fc_sgd_step": (C.c_int, [_P, _P, _P, _P, _F, _F, _I, _F, _I, _P]),
    "fc_client_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _F, _F, _F, _F, _F, _I, _P, _P, _Z, _P]),
    "fc_model_side_stream": (C.c_void_p, [_P]),
    "fc_gather_rows": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "fc_cream_moon_loss": (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _P, _P, _I, _P]),
    "fc_cream_inter_scratch_floats": (C.c_size_t, [_I, _I]),
    "fc_cream_inter_loss": (C.c_int, [_P, _P, _P, _I, _I, _I, _F, _P, _Z, _P, _P, _I, _P]),
    "fc_clip_scratch_bytes": (C.c_size_t, [_P]),
    "fc_clip_grad_norm": (C.c_int, [_P, _P, _F, _P, _Z, _P, _P]),
    "fc_adamw_step_segs": (C.c_int, [_P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P, _I, _P, _P]),
    "fc_mse_loss_fwd_bwd": (C.c_int, [_P, _P, _L, _F, _I, _P, _P, _P]),
    "fc_cream_logprob_diag": (C.c_int, [_P, _P, _I, _I, _P, _P]),
    "fc_cream_combine": (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "fc_client_step_prox": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _F, _F, _F, _F, _F, _I, _P, _P, _Z, _P, _P, _F, _P, _Z]),
    "fc_prox_scratch_bytes": (C.c_size_t, [_P]),
    "fc_prox_term": (C.c_int, [_P, _P, _P, _F, _I, _P, _P, _P, _Z, _P]),
    "fc_aggregate_blend": (C.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "fc_comm_unique_id": (C.c_int, [_P, _Z]),
    "fc_comm_create": (C.c_int, [_P, _Z, _I, _I, C.POINTER(_P)]),
    "fc_comm_destroy": (None, [_P]),
    "fc_comm_rank": (_I, [_P]),
    "fc_comm_world": (_I, [_P]),
    "fc_allreduce_sum": (C.c_int, [_P, _P, _L, _P]),
    "fc_aggregate": (C.c_int, [_P, _P, _P, _L, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "fc_aggregate_partial": (C.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "fc_aggregate_blend_seq": (C.c_int, [_P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "fc_aggregate_exact": (C.c_int, [_P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _P]),
    "fc_copy_outputs": (C.c_int, [_P, _P, _Z, _P, _P, _P]),
    "fc_scale_segments": (C.c_int, [_P, _P, _P, _P, _I, _P]),
    "fc_upload_fold": (C.c_int, [_P, _P, _P, _P]),
    "fc_workspace_tensor": (C.c_int, [_P, _I, _I, _I, _I, C.c_char_p, C.POINTER(_Z), C.POINTER(_Z)]),
    "fc_k_layernorm_fwd": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "fc_k_layernorm_bwd": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "fc_k_layernorm_partial_floats": (_Z, [_I, _I]),
    "fc_k_layernorm_bwd_partial": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "fc_k_gemm": (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "fc_k_gemm_epi": (C.c_int, [_I, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "fc_k_attention_fwd": (C.c_int, [_I, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fc_k_attention_bwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fc_k_dw": (C.c_int, [_I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fc_k_adamw": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "fc_k_cast": (C.c_int, [_I, _P, _P, _L, _P]),
    "fc_image_u8_to_f32": (C.c_int, [_P, _P, _P, _L, _I, _I, _P]),
    "fc_retrieval_scratch_bytes": (C.c_size_t, [_I, _I]),
    "fc_retrieval_best_ranks": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, C.c_size_t, _P, _P]),
    "fc_k_sim_f64": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
}

# the tools build (-DFC_PROBES, FC_PROBES_LIB=1) additionally exports the `#ifdef FC_PROBES` block of the header
PROBES_SIGNATURES = {
    "fc_model_set_option": (C.c_int, [_P, _I, _I]),
    "fc_dbg_step_graph_hits": (C.c_long, [_P]),
    "fc_k_mlp_pack": (C.c_int, [_P, _P, _P, _P, _I, _I, _P]),
    "fc_k_mlp_fused": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
}


def is_probes_build() -> bool:
    return hasattr(lib(), "fc_model_set_option")


def lib():
    """The loaded library.  Raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FedcolaHipError(
                f"{LIB_PATH} not found: build the HIP extension first (python -m fedcola_amd.build). "
                "fedcola_amd has no CPU fallback.")
        import torch  # noqa: F401  (before the library: one HIP runtime per process, the one torch ships -- see __graft_entry__.build)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("FC_LIB_PATH") and not hasattr(l, name):
                continue                      # an older build in an A/B run
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in PROBES_SIGNATURES.items():
            if hasattr(l, name):
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise FedcolaHipError(lib().fc_last_error().decode() or f"libfedcola_hip error {rc}")


def ptr(t):
    """Device (or host) pointer of a torch tensor, or None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
