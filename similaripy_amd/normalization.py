"""Row/column normalisation of sparse matrices.

Only what the similarity hot path needs is here: ``normalize(norm='l1')`` — the first step of
``p3alpha`` / ``rp3beta`` (reference: similaripy/similarity.py:410-415, 477-483 →
similaripy/normalization.py:91-113 → similaripy/cython_code/normalization.pyx:131-161).
``l2`` and ``max`` share the same segmented-reduction shape and are provided as well;
bm25 / bm25plus / tfidf are outside the hot path ("next").
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

_NORMALIZATIONS = ('l1', 'l2', 'max')


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


def _segment_reduce(ufunc, values: np.ndarray, indptr: np.ndarray) -> np.ndarray:
    """ufunc.reduceat over the non-empty rows of a CSR (empty rows -> 0), in values' dtype."""
    n_rows = indptr.shape[0] - 1
    out = np.zeros(n_rows, dtype=values.dtype)
    if values.shape[0]:
        nonempty = np.diff(indptr) > 0
        out[nonempty] = ufunc.reduceat(values, indptr[:-1][nonempty])
    return out


def _inplace_normalize_rows(X: sps.csr_array, norm: str) -> None:
    """Divide every row by its L1 / L2 / max norm; rows whose norm is 0 are left alone
    (normalization.pyx:97-197: `if sum_ == 0.0: continue`).  Arithmetic stays in the data dtype."""
    data, indptr = X.data, X.indptr
    if norm == 'l1':
        norms = _segment_reduce(np.add, np.abs(data), indptr)
    elif norm == 'l2':
        norms = np.sqrt(_segment_reduce(np.add, data * data, indptr))
    else:
        # max of the raw values (not |x|); rows whose max is <= 0 are skipped (normalization.pyx:186-194)
        norms = _segment_reduce(np.maximum, data, indptr)
    norms[norms <= 0] = 1
    data /= np.repeat(norms, np.diff(indptr))


def normalize(X, norm: str = 'l2', axis: int = 1, inplace: bool = False):
    """Normalize a sparse matrix along rows (axis=1) or columns (axis=0) — same signature and
    result as ``similaripy.normalization.normalize`` (normalization.py:91-113)."""
    if norm not in _NORMALIZATIONS:
        raise ValueError(f"norm must be one of {_NORMALIZATIONS}, got '{norm}'")
    X = _prepare_csr(X, axis, inplace)
    _inplace_normalize_rows(X, norm)
    if axis == 0:
        X = X.T
    return X.tocsr()
