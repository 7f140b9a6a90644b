"""ctypes binding of libhsenet_hip.so (the C ABI declared in include/hsenet_hip.h).

There is NO fallback: if the shared object is missing or fails to load, or a call returns a negative
status, this module raises.  No torch types cross the boundary: callers pass ``tensor.data_ptr()`` and sizes.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HS_LIB_PATH") or os.path.join(HERE, "lib", "libhsenet_hip.so")   # override: ablation builds only

HS_OK, HS_ERR_SHAPE, HS_ERR_ARCH, HS_ERR_LAUNCH, HS_ERR_ARG = 0, -1, -2, -3, -4
PREC_BF16, PREC_F32, PREC_F16 = 0, 1, 2
ADAMW_STEP, ADAMW_PREPARE, ADAMW_SWEEP = 0, 1, 2
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_GELU, EPI_GELU_BWD, EPI_RESID, EPI_ATOMIC, EPI_BIAS_M, EPI_SAVE_DGELU = 1, 2, 4, 8, 16, 32, 64
ATTN_Q_PRESCALED, ATTN_F32, ATTN_F16 = 1, 2, 16
LN_DY_F32, LN_DY_BF16, LN_DY_F16, LN_OUT_F16 = 0, 1, 2, 4
Q_PRESCALE = 0.125 * 1.44269504088896341   # dh^-0.5 * log2(e) for dh = 64 (hs_gemm scale_cols_alpha for the q third)

_ERR = {HS_ERR_SHAPE: "unsupported shape/alignment", HS_ERR_ARCH: "no gfx950 device", HS_ERR_LAUNCH: "HIP launch error",
        HS_ERR_ARG: "bad argument"}


class HsError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("batch", C.c_int),
                ("strideA", C.c_int64), ("strideB", C.c_int64), ("strideC", C.c_int64),
                ("out_f32", C.c_void_p), ("out_bf16", C.c_void_p), ("ldc", C.c_int64),
                ("bias", C.c_void_p), ("resid", C.c_void_p), ("aux", C.c_void_p), ("aux_out", C.c_void_p),
                ("colsum_out", C.c_void_p), ("alpha", C.c_float), ("flags", C.c_int), ("layout", C.c_int), ("split_k", C.c_int),
                ("scale_cols", C.c_int), ("scale_cols_alpha", C.c_float)]


ABI_VERSION = 2          # include/hsenet_hip.h HS_ABI_VERSION: load() refuses a library of another ABI


class _Desc(C.Structure):
    """descriptor structs start with `struct_size` (ABI 2): filled in here, so call sites keep passing the reference-facing fields."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class VitDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("B", C.c_int), ("N", C.c_int), ("d", C.c_int), ("H", C.c_int), ("mlp", C.c_int), ("L", C.c_int),
                ("prec", C.c_int), ("save_for_bwd", C.c_int), ("ln_eps", C.c_float), ("qkv_bias", C.c_void_p),
                ("w16", C.c_void_p), ("w16t", C.c_void_p), ("qkv_bias_grad", C.c_void_p)]


class EmbedDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("B", C.c_int), ("Np", C.c_int), ("Pd", C.c_int), ("d", C.c_int), ("has_cls", C.c_int),
                ("scorer", C.c_int), ("ns", C.c_int), ("save_for_bwd", C.c_int), ("drop_p", C.c_float), ("seed", C.c_uint64 * 2),
                ("offset", C.c_uint64 * 2), ("prec", C.c_int)]


class PackerDesc(_Desc):
    _fields_ = [("struct_size", C.c_uint32), ("B", C.c_int), ("gd", C.c_int), ("gh", C.c_int), ("gw", C.c_int), ("kd", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("d", C.c_int), ("out_dim", C.c_int), ("in_row_offset", C.c_int),
                ("in_batch_stride", C.c_int64), ("out_batch_stride", C.c_int64), ("save_for_bwd", C.c_int), ("drop_p", C.c_float),
                ("seed", C.c_uint64 * 2), ("offset", C.c_uint64 * 2), ("prec", C.c_int)]


_vp, _i, _f, _i64, _sz, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t, C.c_uint64

# name -> argtypes (restype int unless listed in _RESTYPE); mirrors include/hsenet_hip.h one to one
SIGNATURES = {
    "hs_version": [],
    "hs_build_id": [],
    "hs_device_check": [_i],
    "hs_gemm": [C.POINTER(GemmArgs), _i, _vp],
    "hs_gemm_group": [C.POINTER(GemmArgs), _i, _vp],
    "hs_gemm_group_prec": [C.POINTER(GemmArgs), _i, _i, _vp],
    "hs_patchify": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "hs_assemble_tokens": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "hs_assemble_tokens_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "hs_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "hs_layernorm_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "hs_attn_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "hs_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "hs_cross_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _u64, _vp],
    "hs_cross_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _u64, _vp],
    "hs_dropout": [_vp, _vp, _i64, _f, _u64, _u64, _vp],
    "hs_rowdot_sigmoid_fwd": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "hs_rowdot_sigmoid_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "hs_pool3d_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "hs_pool3d_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "hs_window_attn_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _u64, _vp],
    "hs_window_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _u64, _vp],
    "hs_colsum": [_vp, _i, _vp, _i, _i, _i64, _vp],
    "hs_cast_bf16": [_vp, _vp, _i64, _vp],
    "hs_cast_f32": [_vp, _vp, _i64, _vp],
    "hs_cast_f16": [_vp, _vp, _i64, _vp],
    "hs_gelu_fwd": [_vp, _vp, _i64, _vp],
    "hs_gelu_bwd": [_vp, _vp, _vp, _i64, _vp],
    "hs_l2norm_fwd": [_vp, _vp, _i, _i, _f, _vp],
    "hs_l2norm_bwd": [_vp, _vp, _vp, _i, _i, _f, _vp],
    "hs_adamw": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _i, _vp],
    "hs_prep_hu_window": [_vp, _i, _vp, _i, _i, _i, C.c_double, C.c_double, C.c_double, C.c_double, _vp],
    "hs_prep_resample": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, C.c_float, _vp],
    "hs_prep_minmax": [_vp, C.c_int64, _vp, _vp],
    "hs_prep_foreground_bbox": [_vp, _i, _i, _i, C.c_float, _vp, _vp],
    "hs_prep_augment": [_vp, _vp, _i, _i, _i, _i, _i, _i, C.c_float, C.c_float, _vp],
    "hs_clip_ce": [_vp, _vp, _vp, _i, _vp],
    "hs_scale": [_vp, _vp, _f, _i64, _vp],
    "hs_prof_enable": [_i],
    "hs_prof_read": [_i, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "hs_prof_name": [_i],
    "hs_prof_count": [],
    "hs_vit_workspace_bytes": [C.POINTER(VitDesc)],
    "hs_vit_weights_elems": [C.POINTER(VitDesc)],
    "hs_vit_cast_weights": [C.POINTER(VitDesc), _vp, _vp, _vp, _vp],
    "hs_vit_blocks_fwd": [C.POINTER(VitDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hs_vit_blocks_fwd_range": [C.POINTER(VitDesc), _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp],
    "hs_vit_blocks_bwd": [C.POINTER(VitDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp],
    "hs_embed_workspace_bytes": [C.POINTER(EmbedDesc)],
    "hs_embed_fwd": [C.POINTER(EmbedDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hs_embed_bwd": [C.POINTER(EmbedDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hs_packer_workspace_bytes": [C.POINTER(PackerDesc)],
    "hs_packer_fwd": [C.POINTER(PackerDesc), _vp, _vp, _vp, _vp, _sz, _vp],
    "hs_packer_bwd": [C.POINTER(PackerDesc), _vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "hs_mse_grad": [_vp, _vp, _vp, _i64, _vp],
    "hs_mse_grad_scaled": [_vp, _vp, _vp, _i64, _f, _vp],
    "hs_sumsq": [_vp, _vp, _vp, _i64, _vp],
    "hs_adamw_masked": [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _vp, _f, _i, _vp],
    "hs_adamw_fused": [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _vp, _f, _i, _vp, _i, _vp, _i, _vp],
    "hs_rows_copy": [_vp, _i64, _vp, _vp, _i64, _i, _i, _i, _vp],
    "hs_rows_scatter": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "hs_scale_dev": [_vp, _vp, _vp, _f, _i64, _vp],
    "hs_dot": [_vp, _vp, _vp, _f, _i64, _vp],
    "hs_rowmix": [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _vp],
    "hs_relation_mse": [_vp, _vp, _vp, _vp, _i64, _f, _i, _vp],
    "hs_ce_rows": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "hs_debug_set": [_i, _i],
    "hs_debug_gemm_stamps": [_vp, _i],
    "hs_debug_gemm_trace": [_vp, _i],
}
_RESTYPE = {"hs_vit_workspace_bytes": _sz, "hs_vit_weights_elems": _sz, "hs_embed_workspace_bytes": _sz, "hs_packer_workspace_bytes": _sz,
            "hs_prof_name": C.c_char_p, "hs_build_id": C.c_char_p, "hs_debug_set": None, "hs_debug_gemm_stamps": None,
            "hs_debug_gemm_trace": None}

_lib = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load (once) and type the shared library.  Raises HsError if it is absent — build it with
    ``python -m hsenet_amd.build`` (or ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HsError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                          "(run `python -m hsenet_amd.build`); hsenet_amd has no CPU/PyTorch fallback")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise HsError(f"cannot load {LIB_PATH}: {e}") from e
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, C.c_int)
        if lib.hs_version() != ABI_VERSION:
            raise HsError(f"{LIB_PATH} speaks ABI {lib.hs_version()}, this package ABI {ABI_VERSION} (include/hsenet_hip.h): rebuild it "
                          "(`python -m hsenet_amd.build`)")
        _lib = lib
        _warn_if_stale(lib)
    return _lib


def _warn_if_stale(lib) -> None:
    """every consumer (tests, bench.py, the modules) comes through load(): say so ONCE when the shared object was not built from the
    sources in this tree (an edited .hip file without a rebuild).  build() / smoke() raise instead (check_fresh); an HS_LIB_PATH
    ablation build and a tree without its csrc/ sources are exempt."""
    if os.environ.get("HS_LIB_PATH"):
        return
    try:
        from .build import source_id
        have, want = lib.hs_build_id().decode(), source_id()
    except Exception:       # noqa: BLE001 - no sources / no hipcc to hash: nothing to compare against
        return
    if have != want:
        import warnings
        warnings.warn(f"{LIB_PATH} is STALE: built from sources {have}, the tree is {want} (run `python -m hsenet_amd.build`)", RuntimeWarning, stacklevel=3)


def build_id() -> str:
    """the content hash the loaded shared object was built from (hsenet_amd/build.py:source_id)."""
    return load().hs_build_id().decode()


def check_fresh() -> str:
    """raise unless the loaded library was built from the sources in this tree (skipped for an HS_LIB_PATH ablation build)."""
    from .build import source_id
    have, want = build_id(), source_id()
    if have != want and not os.environ.get("HS_LIB_PATH"):
        raise HsError(f"{LIB_PATH} is stale: built from sources {have}, the tree is {want} (run `python -m hsenet_amd.build`)")
    return have


def call(name: str, *args) -> None:
    rc = getattr(load(), name)(*args)
    if rc != HS_OK:
        raise HsError(f"{name} failed: {rc} ({_ERR.get(rc, 'unknown')})")
