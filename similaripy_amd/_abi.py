"""ctypes binding of include/sp_knn.h and include/sp_prep.h (libsimilaripy_hip.so).

This is the only way the Python layer reaches compute.  There is no CPU fallback:
if the library is missing, cannot be loaded, or sees no HIP device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from . import _build

SP_SEL_NONE, SP_SEL_ARRAY, SP_SEL_MATRIX = 0, 1, 2
SP_FLAG_TIME_KERNEL = 1
SP_FLAG_NO_ROWS_OUT = 2
SP_FLAG_STATIC_SCHED = 4
SP_FLAG_NO_SPARSE_PATH = 8
SP_FLAG_NO_FOLD = 16
SP_FLAG_NO_ROW_ORDER = 32
SP_FLAG_PHASE_TIMERS = 64
SP_FLAG_M2_IS_M1_T = 128
SP_FLAG_CHECK_ZEROS = 256
SP_FLAG_CSR_OUT = 512
SP_FLAG_P3_PREP = 1024
SP_FLAG_DEPOP_ROWSUM = 2048
SP_FLAG_M1_IS_M2_T = 4096
SP_FLAG_NORMS_ON_DEVICE = 8192
SP_FLAG_REUSE_M2_PREP = 16384
SP_FLAG_BINARY = 32768
SP_FLAG_CHECK_SORTED = 65536
SP_FLAG_PROGRESS = 131072
SP_EZEROS = -6
SP_EUNDERFLOW = -8
SP_EUNSORTED = -7
SP_EUNSORTED_SELECTOR = -9
SP_NORM_L1, SP_NORM_L2, SP_NORM_MAX, SP_NORM_TFIDF, SP_NORM_BM25PLUS = range(5)
SP_TF_MODES = {'binary': 0, 'raw': 1, 'sqrt': 2, 'freq': 3, 'log': 4}       # normalization.pyx:12-17
SP_IDF_MODES = {'unary': 0, 'base': 1, 'smooth': 2, 'prob': 3, 'bm25': 4}   # normalization.pyx:19-24

_c_f32p = C.POINTER(C.c_float)
_c_i32p = C.POINTER(C.c_int32)


class SpKnnArgs(C.Structure):
    """Mirror of ``struct sp_knn_args`` (include/sp_knn.h) — keep field order identical."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("on_device", C.c_int32),
        ("device", C.c_int32),
        ("n_targets", C.c_int32),
        ("n_rows_m1", C.c_int32),
        ("n_rows_m2", C.c_int32),
        ("n_output_cols", C.c_int32),
        ("nnz_m1", C.c_int64),
        ("nnz_m2", C.c_int64),
        ("targets", C.c_void_p),
        ("m1_data", C.c_void_p),
        ("m1_indices", C.c_void_p),
        ("m1_indptr", C.c_void_p),
        ("m2_data", C.c_void_p),
        ("m2_indices", C.c_void_p),
        ("m2_indptr", C.c_void_p),
        ("Xtversky", C.c_void_p),
        ("Ytversky", C.c_void_p),
        ("Xcosine", C.c_void_p),
        ("Ycosine", C.c_void_p),
        ("Xdepop", C.c_void_p),
        ("Ydepop", C.c_void_p),
        ("a1", C.c_float),
        ("l1", C.c_float),
        ("l2", C.c_float),
        ("l3", C.c_float),
        ("t1", C.c_float),
        ("t2", C.c_float),
        ("stabilized_shrink", C.c_float),
        ("bayesian_shrink", C.c_float),
        ("threshold", C.c_float),
        ("k", C.c_int32),
        ("filter_mode", C.c_int32),
        ("filter_m_indptr", C.c_void_p),
        ("filter_m_indices", C.c_void_p),
        ("filter_nnz", C.c_int64),
        ("target_col_mode", C.c_int32),
        ("_pad0", C.c_int32),
        ("target_col_m_indptr", C.c_void_p),
        ("target_col_m_indices", C.c_void_p),
        ("target_col_nnz", C.c_int64),
        ("rows", C.c_void_p),
        ("cols", C.c_void_p),
        ("values", C.c_void_p),
        ("out_counts", C.c_void_p),
        ("stream", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("table_slots", C.c_int32),
        ("threads_per_wg", C.c_int32),
        ("num_wgs", C.c_int32),
        ("load_pct", C.c_int32),
        ("kernel_ms", C.c_float),
        ("passes_total", C.c_int32),
        ("phase_cycles", C.c_int64 * 12),
        ("num_wgs_used", C.c_int32),
        ("_pad1", C.c_int32),
        ("reserved", C.c_int64 * 4),
        ("p3_alpha", C.c_float),
        ("depop_p2", C.c_float),
        ("csr_indptr", C.c_void_p),
        ("csr_nnz", C.c_int64),
        ("explicit_zeros", C.c_int64),
        ("norm_c1", C.c_float), ("norm_c2", C.c_float), ("norm_add", C.c_float), ("_pad2", C.c_int32),
        ("col_keep", C.c_void_p),
        ("n_devices", C.c_int32), ("_pad3", C.c_int32),
        ("device_ids", C.c_void_p),
    ]


# every symbol include/sp_knn.h declares; tests check the library exports all of them
class SpCsrTransposeArgs(C.Structure):
    """Mirror of ``struct sp_csr_transpose_args`` (include/sp_prep.h) — keep field order identical."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("on_device", C.c_int32),
        ("device", C.c_int32),
        ("n_rows", C.c_int32),
        ("n_cols", C.c_int32),
        ("nnz", C.c_int64),
        ("data", C.c_void_p),
        ("indices", C.c_void_p),
        ("indptr", C.c_void_p),
        ("out_data", C.c_void_p),
        ("out_indices", C.c_void_p),
        ("out_indptr", C.c_void_p),
        ("stream", C.c_void_p),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("kernel_ms", C.c_float),
        ("_pad0", C.c_int32),
    ]


class SpCsrSqsumsArgs(C.Structure):
    """Mirror of ``struct sp_csr_sqsums_args`` (include/sp_prep.h) — keep field order identical."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("on_device", C.c_int32),
        ("device", C.c_int32),
        ("n_rows", C.c_int32),
        ("_pad0", C.c_int32),
        ("nnz", C.c_int64),
        ("data", C.c_void_p),
        ("indptr", C.c_void_p),
        ("out_rows", C.c_void_p),
        ("out_cols_of_t", C.c_void_p),
        ("stream", C.c_void_p),
        ("kernel_ms", C.c_float),
        ("_pad1", C.c_int32),
    ]


class SpCsrColsumsArgs(C.Structure):
    """Mirror of ``struct sp_csr_colsums_args`` (include/sp_prep.h) — keep field order identical."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("on_device", C.c_int32),
        ("device", C.c_int32),
        ("n_cols", C.c_int32),
        ("square", C.c_int32),
        ("nnz", C.c_int64),
        ("data", C.c_void_p),
        ("indices", C.c_void_p),
        ("out", C.c_void_p),
        ("stream", C.c_void_p),
        ("kernel_ms", C.c_float),
        ("_pad0", C.c_int32),
    ]


class SpCsrNormalizeArgs(C.Structure):
    """Mirror of ``struct sp_csr_normalize_args`` (include/sp_prep.h) — keep field order identical."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("flags", C.c_uint32),
        ("on_device", C.c_int32),
        ("device", C.c_int32),
        ("n_rows", C.c_int32),
        ("n_cols", C.c_int32),
        ("nnz", C.c_int64),
        ("dtype", C.c_int32),
        ("mode", C.c_int32),
        ("data", C.c_void_p),
        ("indices", C.c_void_p),
        ("indptr", C.c_void_p),
        ("tf_mode", C.c_int32),
        ("idf_mode", C.c_int32),
        ("k1", C.c_double),
        ("b", C.c_double),
        ("delta", C.c_double),
        ("logbase", C.c_double),
        ("pow_alpha", C.c_double),
        ("stream", C.c_void_p),
        ("kernel_ms", C.c_float),
        ("_pad0", C.c_int32),
    ]


EXPORTED_SYMBOLS = (
    "sp_knn_f32_i32",
    "sp_knn_workspace_bytes",
    "sp_device_count",
    "sp_backend_info",
    "sp_last_error",
    "sp_abi_version",
    "sp_csr_transpose_f32_i32",
    "sp_csr_transpose_workspace_bytes",
    "sp_csr_row_sqsums_f32",
    "sp_csr_normalize",
    "sp_csr_col_sums_f32",
    "sp_device_cache_trim",
    "sp_knn_target_costs",
    "sp_knn_partition",
)


class HipLibraryError(RuntimeError):
    pass


class ExplicitZerosError(HipLibraryError):
    """SP_FLAG_CHECK_ZEROS found stored zeros: the caller eliminates them (s_plus.pyx:210-211) and calls again."""


class P3UnderflowError(HipLibraryError):
    """SP_FLAG_P3_PREP: stored entries underflowed to 0.0 in the L1 divide or the power; the reference drops them before its kernel
    runs (similarity.py:410-415, s_plus.pyx:210-211): the caller preprocesses on the host and calls again."""


class UnsortedRowsError(HipLibraryError):
    """SP_FLAG_M1_IS_M2_T found a row of m2 whose column ids do not ascend: the caller converts on the host and calls again."""


class UnsortedSelectorError(UnsortedRowsError):
    """SP_EUNSORTED_SELECTOR: a row of a MATRIX selector does not have ascending column ids (a stale has_sorted_indices flag): the caller
    verifies and sorts a copy of the SELECTOR and calls again."""


_lib = None


def library_path() -> Path:
    return Path(os.environ.get("SIMILARIPY_AMD_LIB", str(_build.LIB_PATH)))


def load(build_if_missing: bool = True):
    """Load (building first if needed) libsimilaripy_hip.so. Raises HipLibraryError on failure."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if build_if_missing and "SIMILARIPY_AMD_LIB" not in os.environ:
        try:
            if _build.is_stale():
                _build.build()
        except Exception as exc:  # no hipcc on the box: use what travelled with the snapshot
            if not path.exists():
                raise HipLibraryError(f"libsimilaripy_hip.so is not built and cannot be built here: {exc}") from exc
    if not path.exists():
        raise HipLibraryError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # torch ships its own libamdhip64.so.7; importing it first makes the HIP runtime a single
    # shared instance, so torch device pointers / streams are valid inside this library.  The plain
    # drop-in path (host buffers in and out) does not need torch: without it the system's HIP runtime is used;
    # device.py / distributed.py import torch themselves.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass

    try:
        lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    except OSError as exc:
        raise HipLibraryError(f"cannot load {path}: {exc}") from exc

    lib.sp_knn_f32_i32.argtypes = [C.POINTER(SpKnnArgs)]
    lib.sp_knn_f32_i32.restype = C.c_int
    lib.sp_knn_workspace_bytes.argtypes = [C.POINTER(SpKnnArgs)]
    lib.sp_knn_workspace_bytes.restype = C.c_int64
    lib.sp_device_count.argtypes = []
    lib.sp_device_count.restype = C.c_int
    lib.sp_backend_info.argtypes = [C.c_int, C.c_char_p, C.c_int]
    lib.sp_backend_info.restype = C.c_int
    lib.sp_last_error.argtypes = []
    lib.sp_last_error.restype = C.c_char_p
    lib.sp_abi_version.argtypes = []
    lib.sp_abi_version.restype = C.c_int
    lib.sp_csr_transpose_f32_i32.argtypes = [C.POINTER(SpCsrTransposeArgs)]
    lib.sp_csr_transpose_f32_i32.restype = C.c_int
    lib.sp_csr_transpose_workspace_bytes.argtypes = [C.POINTER(SpCsrTransposeArgs)]
    lib.sp_csr_transpose_workspace_bytes.restype = C.c_int64
    lib.sp_csr_row_sqsums_f32.argtypes = [C.POINTER(SpCsrSqsumsArgs)]
    lib.sp_csr_row_sqsums_f32.restype = C.c_int
    lib.sp_csr_normalize.argtypes = [C.POINTER(SpCsrNormalizeArgs)]
    lib.sp_csr_normalize.restype = C.c_int
    lib.sp_csr_col_sums_f32.argtypes = [C.POINTER(SpCsrColsumsArgs)]
    lib.sp_csr_col_sums_f32.restype = C.c_int
    lib.sp_device_cache_trim.argtypes = []
    lib.sp_device_cache_trim.restype = C.c_int64
    lib.sp_knn_target_costs.argtypes = [C.POINTER(SpKnnArgs), C.c_void_p]
    lib.sp_knn_target_costs.restype = C.c_int
    lib.sp_knn_partition.argtypes = [C.POINTER(SpKnnArgs), C.c_int, C.c_void_p]
    lib.sp_knn_partition.restype = C.c_int
    _lib = lib
    return lib


def last_error() -> str:
    return load().sp_last_error().decode("utf-8", "replace")


def device_count() -> int:
    return int(load().sp_device_count())


def require_device() -> int:
    n = device_count()
    if n <= 0:
        raise HipLibraryError(
            "no HIP device visible: similaripy_amd runs only on an AMD GPU (gfx950) and has no CPU fallback"
        )
    return n


def backend_info(device: int = 0) -> str:
    buf = C.create_string_buffer(512)
    n = load().sp_backend_info(device, buf, len(buf))
    if n < 0:
        raise HipLibraryError(last_error())
    return buf.value.decode()


def _ptr(a):
    """Address of a numpy array (kept alive by the caller) or 0."""
    if a is None:
        return None
    return a.ctypes.data


def call_knn(args: SpKnnArgs) -> None:
    lib = load()
    args.struct_size = C.sizeof(SpKnnArgs)
    rc = lib.sp_knn_f32_i32(C.byref(args))
    if rc == SP_EZEROS:
        raise ExplicitZerosError(last_error())
    if rc == SP_EUNSORTED_SELECTOR:
        raise UnsortedSelectorError(last_error())
    if rc == SP_EUNSORTED:
        raise UnsortedRowsError(last_error())
    if rc == SP_EUNDERFLOW:
        raise P3UnderflowError(last_error())
    if rc != 0:
        raise HipLibraryError(f"sp_knn_f32_i32 failed ({rc}): {last_error()}")


def call_transpose(args: SpCsrTransposeArgs) -> None:
    lib = load()
    args.struct_size = C.sizeof(SpCsrTransposeArgs)
    rc = lib.sp_csr_transpose_f32_i32(C.byref(args))
    if rc != 0:
        raise HipLibraryError(f"sp_csr_transpose_f32_i32 failed ({rc}): {last_error()}")


def call_row_sqsums(args: SpCsrSqsumsArgs) -> None:
    lib = load()
    args.struct_size = C.sizeof(SpCsrSqsumsArgs)
    rc = lib.sp_csr_row_sqsums_f32(C.byref(args))
    if rc != 0:
        raise HipLibraryError(f"sp_csr_row_sqsums_f32 failed ({rc}): {last_error()}")


def call_normalize(args: SpCsrNormalizeArgs) -> None:
    lib = load()
    args.struct_size = C.sizeof(SpCsrNormalizeArgs)
    rc = lib.sp_csr_normalize(C.byref(args))
    if rc != 0:
        raise HipLibraryError(f"sp_csr_normalize failed ({rc}): {last_error()}")


def call_col_sums(args: SpCsrColsumsArgs) -> None:
    lib = load()
    args.struct_size = C.sizeof(SpCsrColsumsArgs)
    rc = lib.sp_csr_col_sums_f32(C.byref(args))
    if rc != 0:
        raise HipLibraryError(f"sp_csr_col_sums_f32 failed ({rc}): {last_error()}")


def _cost_args(call):
    """sp_knn_args for the partition cost model: the CSR STRUCTURE of the call, its sizes and flags (host pointers; no device needed).
    Returns (args, keep-alive list)."""
    a = SpKnnArgs()
    keep = []

    def i32(x):
        x = as_i32(x); keep.append(x); return x.ctypes.data if x.size else None
    a.on_device = 0
    a.n_targets, a.n_rows_m1, a.n_rows_m2, a.n_output_cols = int(call.targets.shape[0]), call.n_rows_m1, call.n_rows_m2, call.n_output_cols
    a.nnz_m1, a.nnz_m2 = int(call.m1_indices.shape[0]), int(call.m2_indices.shape[0])
    a.flags = (SP_FLAG_M2_IS_M1_T if getattr(call, "m2_is_m1t", False) else 0) | (SP_FLAG_M1_IS_M2_T if getattr(call, "m1_is_m2t", False) else 0)
    a.targets = i32(call.targets)
    a.m1_indices, a.m1_indptr = i32(call.m1_indices), i32(call.m1_indptr)
    a.m2_indices, a.m2_indptr = i32(call.m2_indices), i32(call.m2_indptr)
    a.k = int(call.k)
    a.a1, a.l1, a.l2, a.l3, a.t1, a.t2 = call.a1, call.l1, call.l2, call.l3, call.t1, call.t2
    a.stabilized_shrink, a.bayesian_shrink, a.threshold = call.stabilized_shrink, call.bayesian_shrink, call.threshold
    a.filter_mode, a.target_col_mode = call.filter_mode, call.target_col_mode
    a.struct_size = C.sizeof(SpKnnArgs)
    return a, keep


def target_costs(call) -> np.ndarray:
    """sp_knn_target_costs: what every target slot of the call costs a GPU, in MAC equivalents (the library's ONE partition cost model)."""
    a, keep = _cost_args(call)
    out = np.empty(int(a.n_targets), dtype=np.float64)
    rc = load().sp_knn_target_costs(C.byref(a), out.ctypes.data if out.size else None)
    if rc != 0:
        raise HipLibraryError(f"sp_knn_target_costs failed ({rc}): {last_error()}")
    del keep
    return out


def partition(call, n_parts: int) -> np.ndarray:
    """sp_knn_partition: bounds[0 .. n_parts] of the contiguous cost-balanced slices the library itself cuts `targets` into."""
    a, keep = _cost_args(call)
    out = np.zeros(int(n_parts) + 1, dtype=np.int64)
    rc = load().sp_knn_partition(C.byref(a), int(n_parts), out.ctypes.data)
    if rc != 0:
        raise HipLibraryError(f"sp_knn_partition failed ({rc}): {last_error()}")
    del keep
    return out


def device_cache_trim() -> int:
    """Give the cached device buffers of host-mode calls back to the driver; returns the bytes released."""
    return int(load().sp_device_cache_trim())


def transpose_workspace_bytes(args: SpCsrTransposeArgs) -> int:
    lib = load()
    args.struct_size = C.sizeof(SpCsrTransposeArgs)
    n = lib.sp_csr_transpose_workspace_bytes(C.byref(args))
    if n < 0:
        raise HipLibraryError(f"sp_csr_transpose_workspace_bytes failed ({n}): {last_error()}")
    return int(n)


def workspace_bytes(args: SpKnnArgs) -> int:
    lib = load()
    args.struct_size = C.sizeof(SpKnnArgs)
    n = lib.sp_knn_workspace_bytes(C.byref(args))
    if n < 0:
        raise HipLibraryError(f"sp_knn_workspace_bytes failed ({n}): {last_error()}")
    return int(n)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
