"""Row / column normalisation of sparse matrices, on the GPU.

Same names, arguments, defaults and results as ``similaripy.normalization`` (reference: similaripy/normalization.py:91-218):

    normalize(X, norm='l2'|'l1'|'max', axis, inplace)                       -> normalization.pyx:97-197
    bm25(X, axis, k1, b, logbase, tf_mode, idf_mode, inplace)               -> normalization.pyx:265-334 (delta = 0)
    bm25plus(X, axis, k1, b, delta, logbase, tf_mode, idf_mode, inplace)    -> normalization.pyx:265-334
    tfidf(X, axis, logbase, tf_mode, idf_mode, inplace)                     -> normalization.pyx:200-262

The arithmetic runs in HIP segmented kernels behind ``sp_csr_normalize`` (include/sp_prep.h, csrc/sp_rowops.hpp): one wave
per row, float32 or float64 like the input (other dtypes become float32, normalization.py:23-40).  ``normalize(norm='l1')``
is the first step of ``p3alpha`` / ``rp3beta`` whenever those cannot leave their whole preprocessing to the kernel call
(an explicit ``matrix2``, ``binary``, float64 input).  There is no CPU fallback: without a HIP device the calls raise.
"""
from __future__ import annotations

from math import e

import numpy as np
import scipy.sparse as sps

from . import _abi

_NORMALIZATIONS = ('l1', 'l2', 'max')
_TF_MODES = ('binary', 'raw', 'sqrt', 'freq', 'log')
_IDF_MODES = ('unary', 'base', 'smooth', 'prob', 'bm25')
_NORM_MODE = {'l1': _abi.SP_NORM_L1, 'l2': _abi.SP_NORM_L2, 'max': _abi.SP_NORM_MAX}


def _check_matrix(X):
    # normalization.py:23-40: float32/float64 are kept, anything else becomes float32 CSR
    if not sps.issparse(X):
        raise TypeError("X must be a sparse matrix")
    if X.data.dtype not in (np.float32, np.float64):
        X = sps.csr_array(X, dtype=np.float32)
    return X


def _prepare_csr(X, axis: int, inplace: bool):
    # normalization.py:43-66
    if axis not in (0, 1):
        raise ValueError(f"axis must be 0 or 1, got {axis}")
    X = _check_matrix(X)
    if not inplace:
        X = X.copy()
    if axis == 0:
        X = X.T
    return X.tocsr()


def _finalize_csr(X, axis: int):
    # normalization.py:69-73
    if axis == 0:
        X = X.T
    return X.tocsr()


def _validate_modes(tf_mode: str, idf_mode: str) -> None:
    # normalization.py:76-86
    if tf_mode not in _TF_MODES:
        raise ValueError(f"tf_mode must be one of {_TF_MODES}, got '{tf_mode}'")
    if idf_mode not in _IDF_MODES:
        raise ValueError(f"idf_mode must be one of {_IDF_MODES}, got '{idf_mode}'")


def _device() -> int:
    import os
    return int(os.environ.get("SIMILARIPY_AMD_DEVICE", "0"))


def _run(X: sps.csr_array, mode: int, *, tf_mode: str = 'raw', idf_mode: str = 'unary', k1: float = 0.0, b: float = 0.0,
         delta: float = 0.0, logbase: float = e, pow_alpha: float = 1.0) -> None:
    """In-place weighting of the rows of the CSR `X` on the GPU (host buffers in and out through the C ABI)."""
    _abi.require_device()
    if X.nnz > np.iinfo(np.int32).max:
        raise ValueError("matrix has more than 2^31-1 stored entries (int32 index limit)")
    data = X.data
    if not data.flags.c_contiguous or not data.flags.writeable:
        raise ValueError("X.data must be a writeable contiguous array")
    indices = _abi.as_i32(X.indices)
    indptr = _abi.as_i32(X.indptr)
    a = _abi.SpCsrNormalizeArgs()
    a.on_device = 0
    a.device = _device()
    a.n_rows, a.n_cols, a.nnz = X.shape[0], X.shape[1], int(data.shape[0])
    a.dtype = 1 if data.dtype == np.float64 else 0
    a.mode = mode
    a.data = data.ctypes.data if data.size else None
    a.indices = indices.ctypes.data if indices.size else None
    a.indptr = indptr.ctypes.data
    a.tf_mode, a.idf_mode = _abi.SP_TF_MODES[tf_mode], _abi.SP_IDF_MODES[idf_mode]
    a.k1, a.b, a.delta, a.logbase, a.pow_alpha = float(k1), float(b), float(delta), float(logbase), float(pow_alpha)
    _abi.call_normalize(a)


def normalize(X, norm: str = 'l2', axis: int = 1, inplace: bool = False):
    """Normalize a sparse matrix along rows (axis=1) or columns (axis=0) using L1, L2 or max norm
    (normalization.py:91-113).  Rows whose norm is 0 are left alone."""
    if norm not in _NORMALIZATIONS:
        raise ValueError(f"norm must be one of {_NORMALIZATIONS}, got '{norm}'")
    X = _prepare_csr(X, axis, inplace)
    _run(X, _NORM_MODE[norm])
    return _finalize_csr(X, axis)


def bm25(X, axis: int = 1, k1: float = 1.2, b: float = 0.75, logbase: float = e, tf_mode: str = 'raw',
         idf_mode: str = 'bm25', inplace: bool = False):
    """BM25 weighting (normalization.py:116-148)."""
    _validate_modes(tf_mode, idf_mode)
    X = _prepare_csr(X, axis, inplace)
    _run(X, _abi.SP_NORM_BM25PLUS, tf_mode=tf_mode, idf_mode=idf_mode, k1=k1, b=b, delta=0.0, logbase=logbase)
    return _finalize_csr(X, axis)


def bm25plus(X, axis: int = 1, k1: float = 1.2, b: float = 0.75, delta: float = 1.0, logbase: float = e,
             tf_mode: str = 'raw', idf_mode: str = 'bm25', inplace: bool = False):
    """BM25+ weighting (normalization.py:151-185)."""
    _validate_modes(tf_mode, idf_mode)
    X = _prepare_csr(X, axis, inplace)
    _run(X, _abi.SP_NORM_BM25PLUS, tf_mode=tf_mode, idf_mode=idf_mode, k1=k1, b=b, delta=delta, logbase=logbase)
    return _finalize_csr(X, axis)


def tfidf(X, axis: int = 1, logbase: float = e, tf_mode: str = 'sqrt', idf_mode: str = 'smooth', inplace: bool = False):
    """TF-IDF weighting (normalization.py:188-218)."""
    _validate_modes(tf_mode, idf_mode)
    X = _prepare_csr(X, axis, inplace)
    _run(X, _abi.SP_NORM_TFIDF, tf_mode=tf_mode, idf_mode=idf_mode, logbase=logbase)
    return _finalize_csr(X, axis)
