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

_NORM_MODE = {'l1': _abi.SP_NORM_L1, 'l2': _abi.SP_NORM_L2, 'max': _abi.SP_NORM_MAX}


def _one_of(name: str, value, allowed) -> None:
    """The reference's messages for a bad mode string (normalization.py:76-86, :106-107)."""
    if value not in allowed:
        raise ValueError(f"{name} must be one of {tuple(allowed)}, got '{value}'")


def _rows_view(X, axis: int, inplace: bool):
    """The CSR whose ROWS are the vectors to weight, and the way back to X's orientation.

    What the reference does around its Cython kernels (normalization.py:23-73): float32 / float64 data are kept, any other
    dtype becomes float32; `inplace=False` works on a copy; axis=0 weights the columns by transposing, which always
    re-materialises the matrix (so `inplace` only reaches the caller's buffer for a CSR weighted along axis=1)."""
    if axis not in (0, 1):
        raise ValueError(f"axis must be 0 or 1, got {axis}")
    if not sps.issparse(X):
        raise TypeError("X must be a sparse matrix")
    work = X if X.data.dtype in (np.float32, np.float64) else sps.csr_array(X, dtype=np.float32)
    if not inplace:
        work = work.copy()
    if axis == 1:
        return work.tocsr(), (lambda rows: rows)
    return work.T.tocsr(), (lambda rows: rows.T.tocsr())


def _weighted(X, axis: int, inplace: bool, mode: int, **params):
    rows, back = _rows_view(X, axis, inplace)
    _run(rows, mode, **params)
    return back(rows)


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
    _one_of("norm", norm, _NORM_MODE)
    return _weighted(X, axis, inplace, _NORM_MODE[norm])


def bm25(X, axis: int = 1, k1: float = 1.2, b: float = 0.75, logbase: float = e, tf_mode: str = 'raw',
         idf_mode: str = 'bm25', inplace: bool = False):
    """BM25 weighting (normalization.py:116-148): BM25+ with delta = 0."""
    return bm25plus(X, axis=axis, k1=k1, b=b, delta=0.0, logbase=logbase, tf_mode=tf_mode, idf_mode=idf_mode, inplace=inplace)


def bm25plus(X, axis: int = 1, k1: float = 1.2, b: float = 0.75, delta: float = 1.0, logbase: float = e,
             tf_mode: str = 'raw', idf_mode: str = 'bm25', inplace: bool = False):
    """BM25+ weighting (normalization.py:151-185)."""
    _one_of("tf_mode", tf_mode, _abi.SP_TF_MODES)
    _one_of("idf_mode", idf_mode, _abi.SP_IDF_MODES)
    return _weighted(X, axis, inplace, _abi.SP_NORM_BM25PLUS, tf_mode=tf_mode, idf_mode=idf_mode, k1=k1, b=b, delta=delta, logbase=logbase)


def tfidf(X, axis: int = 1, logbase: float = e, tf_mode: str = 'sqrt', idf_mode: str = 'smooth', inplace: bool = False):
    """TF-IDF weighting (normalization.py:188-218)."""
    _one_of("tf_mode", tf_mode, _abi.SP_TF_MODES)
    _one_of("idf_mode", idf_mode, _abi.SP_IDF_MODES)
    return _weighted(X, axis, inplace, _abi.SP_NORM_TFIDF, tf_mode=tf_mode, idf_mode=idf_mode, logbase=logbase)
