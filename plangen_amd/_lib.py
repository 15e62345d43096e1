"""ctypes binding of include/plangen_hip.h.  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libplangen_hip.so")
# Diagnostics superset (same object files + microbenchmarks / hazard screens / kernel-variant tables and the switches that make a handle
# compute something else): loaded only by tools/, bench.py's instrumented pass and a few GPU tests -- never by the product path.
DIAG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libplangen_diag.so")

PG_F32, PG_BF16, PG_I32, PG_I64 = 0, 1, 2, 3
PG_MAX_VQ_LEVELS = 8
STATUS = {0: "PG_OK", -1: "PG_ERR_ARG", -2: "PG_ERR_HIP", -3: "PG_ERR_STATE", -4: "PG_ERR_NAME", -5: "PG_ERR_CAPACITY"}


class pg_config(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("inter", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
        ("head_dim", C.c_int32), ("vocab", C.c_int32),
        ("img_vocab", C.c_int32), ("img_dim", C.c_int32), ("grid", C.c_int32), ("gen_head_dim", C.c_int32),
        ("vq_ch", C.c_int32), ("vq_levels", C.c_int32), ("vq_ch_mult", C.c_int32 * PG_MAX_VQ_LEVELS),
        ("vq_z", C.c_int32), ("vq_res_blocks", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float),
        ("compute_dtype", C.c_int32), ("max_rows", C.c_int32), ("max_prompt", C.c_int32),
        ("max_new", C.c_int32), ("max_images", C.c_int32), ("with_lm_head", C.c_int32),
        ("with_vq_encoder", C.c_int32),
        ("with_vision", C.c_int32), ("vit_width", C.c_int32), ("vit_layers", C.c_int32), ("vit_heads", C.c_int32),
        ("vit_mlp", C.c_int32), ("vit_patch", C.c_int32), ("vit_img", C.c_int32), ("max_vision_images", C.c_int32),
    ]


class pg_timing(C.Structure):
    _fields_ = [("decode_ms", C.c_float), ("attn_ms_sum", C.c_float), ("attn_launches", C.c_int32),
                ("attn_bytes_sum", C.c_double), ("prefill_ms", C.c_float), ("vq_ms", C.c_float)]


# every symbol include/plangen_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("pg_create", C.c_int, [C.POINTER(_P), C.POINTER(pg_config), C.c_int]),
    ("pg_destroy", C.c_int, [_P]),
    ("pg_last_error", C.c_char_p, [_P]),
    ("pg_load_tensor", C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    ("pg_finalize_weights", C.c_int, [_P, C.POINTER(C.c_int), _P]),
    ("pg_prefill", C.c_int, [_P, _P, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    ("pg_prefill_embeds", C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    ("pg_step", C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P]),
    ("pg_gen_head", C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P]),
    ("pg_gen_embed", C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    ("pg_embed_tokens", C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    ("pg_decode_image_tokens", C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_uint64, _P, _P, _P, _P, _P]),
    ("pg_generate_text_greedy", C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_int), _P]),
    ("pg_vq_decode", C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    ("pg_vq_encode", C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P]),
    ("pg_vision_encode", C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, _P]),
    ("pg_get_timing", C.c_int, [_P, C.POINTER(pg_timing)]),
    ("pg_get_class_timing", C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("pg_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("pg_device_bytes", C.c_int64, [_P]),
    ("pg_debug_read", C.c_int, [_P, C.c_char_p, C.c_int, _P, C.c_int64, _P]),
    ("pg_op_rmsnorm", C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    ("pg_op_gemm", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _P]),
    ("pg_op_swiglu_gemm", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("pg_op_uniform", C.c_int, [_P, _P, _P, C.c_int, _P]),
    ("pg_op_conv3x3", C.c_int, [_P, _P, _P, _P, _P, _P] + [C.c_int] * 7 + [_P]),
    ("pg_op_groupnorm", C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
]

_lib = None
_diag = None


def _bind(lib: C.CDLL) -> C.CDLL:
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def load_diag() -> C.CDLL:
    """Load libplangen_diag.so: every symbol of the product ABI (a handle created through it runs the product's own object code) plus
    ``pg_diag_set_option`` and the ``pg_bench_*`` measurement entry points.  Measurement / test infrastructure only."""
    global _diag
    if _diag is not None:
        return _diag
    import torch  # noqa: F401  (same reason as in load())
    if not os.path.exists(DIAG_LIB_PATH):
        raise RuntimeError(f"plangen_amd: diagnostics library not found at {DIAG_LIB_PATH}; build it with `make -C plangen_amd/csrc`.")
    lib = _bind(C.CDLL(DIAG_LIB_PATH))
    lib.pg_diag_set_option.restype = C.c_int
    lib.pg_diag_set_option.argtypes = [_P, C.c_char_p, C.c_int64]
    _diag = lib
    return lib


def load() -> C.CDLL:
    """Load libplangen_hip.so (built by ``__graft_entry__.build()`` / ``make -C plangen_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime; it must be the one already mapped when our library's
    # libamdhip64 dependency is resolved, or the process ends up with two runtimes (and ours
    # sees no device).  PyTorch is the allocator / stream provider anyway.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"plangen_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C plangen_amd/csrc`. "
            "There is no CPU fallback.")
    lib = _bind(C.CDLL(LIB_PATH))
    _lib = lib
    return lib
