"""ctypes binding of libvispec_hip.so (include/vispec_hip.h).  There is NO fallback: if the HIP library cannot be
loaded the product path raises — nothing in vispec_amd ever routes through the oracle or a CPU/eager substitute."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvispec_hip.so")
if os.environ.get("VISPEC_LIB_VARIANT"):  # same-box A/Bs of compile-time switches (tools/): another in-tree build of the SAME sources, e.g. libvispec_hip_noswz.so
    LIB_PATH = os.path.join(_HERE, f"libvispec_hip_{os.environ['VISPEC_LIB_VARIANT']}.so")

c_void_p, c_int, c_float = C.c_void_p, C.c_int, C.c_float


class VispecConfig(C.Structure):
    _fields_ = [
        ("hidden_size", c_int), ("num_heads", c_int), ("num_kv_heads", c_int), ("head_dim", c_int),
        ("intermediate_size", c_int), ("vocab_size", c_int), ("num_layers", c_int), ("max_pos", c_int),
        ("rms_eps", c_float), ("qkv_bias", c_int),
        ("draft_heads", c_int), ("draft_intermediate", c_int), ("draft_max_pos", c_int), ("draft_qkv_bias", c_int),
        ("draft_fc_bias", c_int), ("draft_rms_eps", c_float),
        ("total_token", c_int), ("depth", c_int), ("top_k", c_int), ("num_q", c_int),
        ("eos_token_id", c_int), ("eager_scores", c_int), ("draft_rope_rows", c_int),
    ]


class LayerWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("wqkv", "bqkv", "wo", "wgu", "wdown", "ln1", "ln2", "sqkv", "so", "sgu", "sdown")]


class TargetMisc(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("embed", "norm", "lm_head", "lm_head_scale", "rope_cos", "rope_sin")]


class DraftWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("embed", "fc_w", "fc_b", "imgfc_w", "imgfc_b", "wqkv", "bqkv", "wo", "wgu", "wdown",
                                        "ln2", "ad_q", "ad_wkv", "ad_bkv", "ad_wo", "rope_cos", "rope_sin")]


# every symbol include/vispec_hip.h declares: name -> (restype, argtypes)
P = c_void_p
SIGNATURES = {
    "vispec_last_error": (C.c_char_p, []),
    "vispec_version": (c_int, []),
    "vispec_ctx_create": (c_int, [C.POINTER(VispecConfig), C.POINTER(P)]),
    "vispec_ctx_create_member": (c_int, [C.POINTER(VispecConfig), P, C.POINTER(P)]),
    "vispec_ctx_destroy": (None, [P]),
    "vispec_set_target_layer": (c_int, [P, c_int, C.POINTER(LayerWeights)]),
    "vispec_set_target_misc": (c_int, [P, C.POINTER(TargetMisc)]),
    "vispec_set_draft_weights": (c_int, [P, C.POINTER(DraftWeights)]),
    "vispec_set_kv": (c_int, [P, P, P]),
    "vispec_gemm_skinny": (c_int, [P, P, P, c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int]),
    "vispec_pack_weight": (c_int, [P, P, P, c_int, c_int, P]),
    "vispec_packed_elems": (C.c_longlong, [c_int, c_int]),
    "vispec_pack_weight_fp8": (c_int, [P, P, P, c_int, c_int, P]),
    "vispec_gemm_skinny_fp8": (c_int, [P, P, P, c_int, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int]),
    "vispec_gemm_skinny_norm": (c_int, [P, P, P, c_int, P, P, P, c_int, P, c_int, P, P, c_int, c_float, c_int, c_int, c_int]),
    "vispec_gemm_skinny_tune": (c_int, [P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int]),
    "vispec_rmsnorm": (c_int, [P, P, P, P, P, c_int, c_int, c_float]),
    "vispec_set_total_token": (c_int, [P, c_int]),
    "vispec_set_top_k": (c_int, [P, c_int]),
    "vispec_set_stop_token": (c_int, [P, P, c_int]),
    "vispec_qkv_rope_fused": (c_int, [c_int]),
    "vispec_gemm_qkv_rope": (c_int, [P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P]),
    "vispec_silu_mul": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int]),
    "vispec_scale_bias_cast": (c_int, [P, P, P, c_int, P, P, P, c_int, c_int, c_int]),
    "vispec_add_rmsnorm": (c_int, [P, P, P, P, P, P, c_int, c_int, c_float]),
    "vispec_prefill_attention": (c_int, [P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, P, c_int, c_int]),
    "vispec_rope_append": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P]),
    "vispec_tree_attention": (c_int, [P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, c_int]),
    "vispec_argmax_rows": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "vispec_logsoftmax_topk": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "vispec_begin_request": (c_int, [P, P, P, c_int, c_int]),
    "vispec_draft_prefill": (c_int, [P, P, P, P, P, c_int, P]),
    "vispec_verify_accept": (c_int, [P, P, c_int]),
    "vispec_target_forward": (c_int, [P, P]),
    "vispec_accept": (c_int, [P, P, c_int]),
    "vispec_set_tree_host": (c_int, [P, P, P, P, P, P, c_int, c_int]),
    "vispec_set_retrieve_host": (c_int, [P, P, P, c_int, c_int]),
    "vispec_draft_round": (c_int, [P, P]),
    "vispec_cohort_verify_accept": (c_int, [P, P, P, c_int]),
    "vispec_cohort_draft_round": (c_int, [P, P, P]),
    "vispec_cohortn_verify_accept": (c_int, [P, c_int, P, c_int]),
    "vispec_cohortn_draft_round": (c_int, [P, c_int, P]),
    "vispec_gemm_fp8a8": (c_int, [P, P, P, c_int, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float]),
    "vispec_a8_scratch_read": (c_int, [P, P, P, P, c_int, c_int]),
    "vispec_quant_rows_e4m3": (c_int, [P, P, P, c_int, P, c_int, P, c_int, c_int]),
    "vispec_gemm_cohort": (c_int, [P, P, P, c_int, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int]),
    "vispec_set_rope_delta": (c_int, [P, P, c_int]),
    "vispec_set_sampling": (c_int, [P, c_float, C.c_ulonglong]),
    "vispec_sample_row": (c_int, [P, P, P, c_int, P]),
    "vispec_set_uniform_override_host": (c_int, [P, P, P, c_int, c_int, c_float]),
    "vispec_set_next_token": (c_int, [P, P, P]),
    "vispec_ar_step": (c_int, [P, P]),
    "vispec_cohortn_ar_step": (c_int, [P, c_int, P]),
    "vispec_get_state_host": (c_int, [P, P, P]),
    "vispec_cohort_get_state_host": (c_int, [P, c_int, P, P]),
    "vispec_cohort_state_enqueue": (c_int, [P, c_int, P, c_int]),
    "vispec_cohort_state_wait": (c_int, [P, c_int, c_int, P]),
    "vispec_get_last_accept_host": (c_int, [P, P, P]),
    "vispec_get_tokens_host": (c_int, [P, P, P, c_int]),
    "vispec_get_accept_log_host": (c_int, [P, P, P, c_int]),
    "vispec_get_tree_host": (c_int, [P, P, P, P, P, P, P, P]),
    "vispec_set_graphs": (c_int, [P, c_int]),
    "vispec_set_fp8_activations": (c_int, [P, c_int]),
    "vispec_set_wide_row_blocks": (c_int, [P, c_int]),
    "vispec_graph_stats": (c_int, [P, P]),
    "vispec_prof_enable": (c_int, [P, c_int]),
    "vispec_prof_report_host": (c_int, [P, P, P, c_int]),
    "vispec_buffer": (P, [P, C.c_char_p]),
}

TREE_MAX_T = 64
TREE_RET_W = 10

_lib = None


class VispecError(RuntimeError):
    pass


def load(build_if_missing: bool = False) -> C.CDLL:
    """Load the shared library (optionally building it first).  Raises if unavailable — by design."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64; loading ours before it would bring up a second, device-less HIP runtime
    import torch  # noqa: F401
    if build_if_missing:
        from . import build as _b
        _b.build(verbose=False)
    if not os.path.exists(LIB_PATH):
        raise VispecError(f"{LIB_PATH} is missing: run `python -m vispec_amd.build` (hipcc --offload-arch=gfx950). "
                          "There is no CPU/eager fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise VispecError((load().vispec_last_error() or b"?").decode())
