"""ORACLE — TEST INFRASTRUCTURE ONLY.  NumPy restatement of the reference's row normalisers

    similaripy/normalization.py:91-218                    normalize / bm25 / bm25plus / tfidf (argument handling)
    similaripy/cython_code/normalization.pyx:97-197       l1 / l2 / max
    similaripy/cython_code/normalization.pyx:200-262      tf-idf
    similaripy/cython_code/normalization.pyx:265-334      BM25+ (BM25 = delta 0)

used by tests/ to check the device kernels behind similaripy_amd.normalization (sp_csr_normalize, include/sp_prep.h), and
pinned itself by golden vectors generated from the imported reference (tests/golden/make_norm_golden.py).
Arithmetic follows the Cython code in the data type (float32 stays float32); row sums are np.add.reduceat (the reference's
loops are sequential sums that its compiler reorders under -ffast-math: equal to a few ulp either way).
Nothing under similaripy_amd/ imports this module.
"""
from __future__ import annotations

from math import e

import numpy as np
import scipy.sparse as sps

_NORMALIZATIONS = ('l1', 'l2', 'max')
_TF_MODES = ('binary', 'raw', 'sqrt', 'freq', 'log')
_IDF_MODES = ('unary', 'base', 'smooth', 'prob', 'bm25')


def _check_matrix(X):
    if not sps.issparse(X):
        raise TypeError("X must be a sparse matrix")
    if X.data.dtype not in (np.float32, np.float64):
        X = sps.csr_array(X, dtype=np.float32)
    return X


def _prepare_csr(X, axis: int, inplace: bool):
    if axis not in (0, 1):
        raise ValueError(f"axis must be 0 or 1, got {axis}")
    X = _check_matrix(X)
    if not inplace:
        X = X.copy()
    if axis == 0:
        X = X.T
    return X.tocsr()


def _finalize(X, axis):
    if axis == 0:
        X = X.T
    return X.tocsr()


def _segment_reduce(ufunc, values: np.ndarray, indptr: np.ndarray) -> np.ndarray:
    n_rows = indptr.shape[0] - 1
    out = np.zeros(n_rows, dtype=values.dtype)
    if values.shape[0]:
        nonempty = np.diff(indptr) > 0
        out[nonempty] = ufunc.reduceat(values, indptr[:-1][nonempty])
    return out


def normalize(X, norm: str = 'l2', axis: int = 1, inplace: bool = False):
    if norm not in _NORMALIZATIONS:
        raise ValueError(f"norm must be one of {_NORMALIZATIONS}, got '{norm}'")
    X = _prepare_csr(X, axis, inplace)
    data, indptr = X.data, X.indptr
    if norm == 'l1':
        norms = _segment_reduce(np.add, np.abs(data), indptr)
    elif norm == 'l2':
        norms = np.sqrt(_segment_reduce(np.add, data * data, indptr))
    else:
        norms = _segment_reduce(np.maximum, data, indptr)      # max of the raw values; rows whose max is <= 0 are skipped
    norms[norms <= 0] = 1
    data /= np.repeat(norms, np.diff(indptr))
    return _finalize(X, axis)


def _tf(freq, doc_len, mode, llb):
    T = freq.dtype.type
    if mode == 'binary':
        return (freq != 0).astype(freq.dtype)
    if mode == 'raw':
        return freq
    if mode == 'sqrt':
        return np.sqrt(freq)
    if mode == 'freq':
        return freq / doc_len
    return (np.log(1.0 + freq.astype(np.float64)) / np.float64(llb)).astype(freq.dtype)      # (1 + freq: Cython promotes int + float to double)


def _idf(df, n_docs, mode, llb, dtype):
    T = np.dtype(dtype).type
    f = df.astype(dtype)
    n = T(n_docs)
    with np.errstate(divide='ignore', invalid='ignore'):
        if mode == 'unary':
            v = np.ones_like(f)
        elif mode == 'base':
            v = (np.log((n / f).astype(np.float64)) / np.float64(llb)).astype(dtype)
        elif mode == 'smooth':
            v = (np.log(np.float64(n) / (1.0 + f.astype(np.float64))) / np.float64(llb)).astype(dtype)
        elif mode == 'prob':
            v = (np.log(((n - f) / f).astype(np.float64)) / np.float64(llb)).astype(dtype)
        else:
            v = (np.log(((n - f).astype(np.float64) + 0.5) / (f.astype(np.float64) + 0.5)) / np.float64(llb)).astype(dtype)
    v[df == 0] = 0
    return v


def _weighted(X, axis, inplace, tf_mode, idf_mode, logbase, k1=None, b=None, delta=None):
    if tf_mode not in _TF_MODES:
        raise ValueError(f"tf_mode must be one of {_TF_MODES}, got '{tf_mode}'")
    if idf_mode not in _IDF_MODES:
        raise ValueError(f"idf_mode must be one of {_IDF_MODES}, got '{idf_mode}'")
    X = _prepare_csr(X, axis, inplace)
    data, indices, indptr = X.data, X.indices, X.indptr
    dt = data.dtype
    T = dt.type
    n_docs, n_words = X.shape
    llb = T(np.log(logbase))
    doc_len = _segment_reduce(np.add, data, indptr)
    df = np.bincount(indices[data > 0], minlength=n_words)
    idf = _idf(df, n_docs, idf_mode, llb, dt)
    rows = np.repeat(np.arange(n_docs), np.diff(indptr))
    tf = _tf(data, doc_len[rows], tf_mode, llb)
    if k1 is None:
        data[:] = tf * idf[indices]
    elif n_docs:
        avg = T(np.add.reduce(doc_len, dtype=dt) / T(n_docs))
        k1, b, delta = T(k1), T(b), T(delta)
        norm_len = ((1.0 - np.float64(b)) + (b * doc_len / avg).astype(np.float64)).astype(dt)
        den = tf + k1 * norm_len[rows]
        data[:] = (idf[indices].astype(np.float64) * ((tf.astype(np.float64) * (np.float64(k1) + 1.0)) / den.astype(np.float64) + np.float64(delta))).astype(dt)
    return _finalize(X, axis)


def bm25(X, axis=1, k1=1.2, b=0.75, logbase=e, tf_mode='raw', idf_mode='bm25', inplace=False):
    return _weighted(X, axis, inplace, tf_mode, idf_mode, logbase, k1, b, 0.0)


def bm25plus(X, axis=1, k1=1.2, b=0.75, delta=1.0, logbase=e, tf_mode='raw', idf_mode='bm25', inplace=False):
    return _weighted(X, axis, inplace, tf_mode, idf_mode, logbase, k1, b, delta)


def tfidf(X, axis=1, logbase=e, tf_mode='sqrt', idf_mode='smooth', inplace=False):
    return _weighted(X, axis, inplace, tf_mode, idf_mode, logbase)


def inplace_run(X, mode, *, tf_mode='raw', idf_mode='unary', k1=0.0, b=0.0, delta=0.0, logbase=e, pow_alpha=1.0):
    """Stand-in with the signature of similaripy_amd.normalization._run (in-place weighting of the rows of the CSR X) for the
    CPU test tier, where the product's device call cannot run: tests/conftest.py patches it in to pin the HOST logic of the
    wrappers against the reference's golden vectors."""
    names = {0: 'l1', 1: 'l2', 2: 'max'}
    if mode in names:
        Y = normalize(X, norm=names[mode])
        X.data[:] = Y.data if pow_alpha == 1.0 else np.power(Y.data, X.data.dtype.type(pow_alpha))
    elif mode == 3:
        X.data[:] = tfidf(X, logbase=logbase, tf_mode=tf_mode, idf_mode=idf_mode).data
    else:
        X.data[:] = bm25plus(X, k1=k1, b=b, delta=delta, logbase=logbase, tf_mode=tf_mode, idf_mode=idf_mode).data
