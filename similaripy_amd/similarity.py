"""Public similarity functions — the drop-in surface.

Same names, positional order, keyword names and defaults as ``similaripy.similarity``
(reference: similaripy/similarity.py:9-617).  Every function is one parameterisation of the
single GPU kernel behind ``_host.s_plus``:

    dot_product        :9    no normalisation
    cosine             :67   l2=1, c1=c2=0.5
    asymmetric_cosine  :126  l2=1, c1=alpha, c2=1-alpha
    tversky            :189  l1=1, t1=alpha, t2=beta
    jaccard            :252  l1=1, t1=t2=1
    dice               :311  l1=1, t1=t2=0.5
    p3alpha            :370  rows of m1 and m2 L1-normalised, then data**alpha
    rp3beta            :435  as p3alpha + l3=1, depop weights = column sums of raw m2, p2=beta
    s_plus             :506  everything exposed

All return a ``scipy.sparse`` ``coo_array`` (default) or ``csr_array`` of float32 with shape
``(matrix1.shape[0], matrix2.shape[1])``.
"""
from __future__ import annotations

from typing import Literal, Optional, Union

import numpy as np
from scipy.sparse import sparray

from . import _abi, _host
from .normalization import normalize as _normalize

_Rows = Optional[Union[list, np.ndarray]]
_Cols = Optional[Union[list, np.ndarray, sparray]]
_Shrink = Literal['stabilized', 'bayesian', 'additive']
_Fmt = Literal['csr', 'coo']


def __get_shrink_values__(shrink: float, shrink_type: str) -> tuple:
    """shrink -> (stabilized, bayesian, additive) — similarity.py:595-617."""
    if shrink_type == 'stabilized':
        return shrink, 0.0, 0.0
    if shrink_type == 'bayesian':
        return 0.0, shrink, 0.0
    if shrink_type == 'additive':
        return 0.0, 0.0, shrink
    raise ValueError("shrink_type must be one of 'stabilized', 'bayesian', or 'additive'")


def _run(matrix1, matrix2, kernel_kw, k, shrink, shrink_type, threshold, binary, target_rows, target_cols,
         filter_cols, verbose, format_output, num_threads, block_size):
    stab, bayes, add = __get_shrink_values__(shrink, shrink_type)
    return _host.s_plus(
        matrix1, matrix2=matrix2, k=k,
        stabilized_shrink=stab, bayesian_shrink=bayes, additive_shrink=add,
        threshold=threshold, binary=binary,
        target_rows=target_rows, target_cols=target_cols, filter_cols=filter_cols,
        verbose=verbose, format_output=format_output, num_threads=num_threads, block_size=block_size,
        **kernel_kw)


def dot_product(matrix1: sparray, matrix2: Optional[sparray] = None, k: int = 100, shrink: float = 0.0,
                shrink_type: _Shrink = 'stabilized', threshold: float = 0.0, binary: bool = False,
                target_rows: _Rows = None, target_cols: _Cols = None, filter_cols: _Cols = None,
                verbose: bool = True, format_output: _Fmt = 'coo', num_threads: int = 0,
                block_size: Optional[int] = 0) -> sparray:
    """Top-k dot product between rows of matrix1 and columns of matrix2 (default matrix1.T)."""
    return _run(matrix1, matrix2, {}, k, shrink, shrink_type, threshold, binary, target_rows, target_cols,
                filter_cols, verbose, format_output, num_threads, block_size)


def cosine(matrix1: sparray, matrix2: Optional[sparray] = None, k: int = 100, shrink: float = 0.0,
           shrink_type: _Shrink = 'stabilized', threshold: float = 0.0, binary: bool = False,
           target_rows: _Rows = None, target_cols: _Cols = None, filter_cols: _Cols = None,
           verbose: bool = True, format_output: _Fmt = 'coo', num_threads: int = 0,
           block_size: Optional[int] = 0) -> sparray:
    """Top-k cosine similarity: xy / (|x| |y| + shrink)."""
    return _run(matrix1, matrix2, dict(l2=1, c1=0.5, c2=0.5), k, shrink, shrink_type, threshold, binary,
                target_rows, target_cols, filter_cols, verbose, format_output, num_threads, block_size)


def asymmetric_cosine(matrix1: sparray, matrix2: Optional[sparray] = None, alpha: float = 0.5, k: int = 100,
                      shrink: float = 0.0, shrink_type: _Shrink = 'stabilized', threshold: float = 0.0,
                      binary: bool = False, target_rows: _Rows = None, target_cols: _Cols = None,
                      filter_cols: _Cols = None, verbose: bool = True, format_output: _Fmt = 'coo',
                      num_threads: int = 0, block_size: Optional[int] = 0) -> sparray:
    """Top-k asymmetric cosine: xy / (|x|^(2 alpha) |y|^(2 (1-alpha)) + shrink)."""
    return _run(matrix1, matrix2, dict(l2=1, c1=alpha, c2=1 - alpha), k, shrink, shrink_type, threshold, binary,
                target_rows, target_cols, filter_cols, verbose, format_output, num_threads, block_size)


def tversky(matrix1: sparray, matrix2: Optional[sparray] = None, alpha: float = 1.0, beta: float = 1.0,
            k: int = 100, shrink: float = 0.0, shrink_type: _Shrink = 'stabilized', threshold: float = 0.0,
            binary: bool = False, target_rows: _Rows = None, target_cols: _Cols = None,
            filter_cols: _Cols = None, verbose: bool = True, format_output: _Fmt = 'coo',
            num_threads: int = 0, block_size: Optional[int] = 0) -> sparray:
    """Top-k Tversky index: xy / (alpha (|x|^2 - xy) + beta (|y|^2 - xy) + xy + shrink)."""
    return _run(matrix1, matrix2, dict(l1=1, t1=alpha, t2=beta), k, shrink, shrink_type, threshold, binary,
                target_rows, target_cols, filter_cols, verbose, format_output, num_threads, block_size)


def jaccard(matrix1: sparray, matrix2: Optional[sparray] = None, k: int = 100, shrink: float = 0.0,
            shrink_type: _Shrink = 'stabilized', threshold: float = 0.0, binary: bool = False,
            target_rows: _Rows = None, target_cols: _Cols = None, filter_cols: _Cols = None,
            verbose: bool = True, format_output: _Fmt = 'coo', num_threads: int = 0,
            block_size: Optional[int] = 0) -> sparray:
    """Top-k Jaccard (Tversky with alpha = beta = 1)."""
    return _run(matrix1, matrix2, dict(l1=1, t1=1, t2=1), k, shrink, shrink_type, threshold, binary,
                target_rows, target_cols, filter_cols, verbose, format_output, num_threads, block_size)


def dice(matrix1: sparray, matrix2: Optional[sparray] = None, k: int = 100, shrink: float = 0.0,
         shrink_type: _Shrink = 'stabilized', threshold: float = 0.0, binary: bool = False,
         target_rows: _Rows = None, target_cols: _Cols = None, filter_cols: _Cols = None,
         verbose: bool = True, format_output: _Fmt = 'coo', num_threads: int = 0,
         block_size: Optional[int] = 0) -> sparray:
    """Top-k Dice (Tversky with alpha = beta = 0.5)."""
    return _run(matrix1, matrix2, dict(l1=1, t1=0.5, t2=0.5), k, shrink, shrink_type, threshold, binary,
                target_rows, target_cols, filter_cols, verbose, format_output, num_threads, block_size)


def _p3_inputs(matrix1, matrix2, alpha):
    # similarity.py:408-415: rows of m1 AND rows of m2 divided by their L1 norm, then ^alpha
    if matrix2 is None:
        matrix2 = matrix1.T
    raw_m2 = matrix2
    matrix1 = _normalize(matrix1, norm='l1', axis=1, inplace=False)
    matrix1.data = np.power(matrix1.data, alpha)
    matrix2 = _normalize(matrix2, norm='l1', axis=1, inplace=False)
    matrix2.data = np.power(matrix2.data, alpha)
    return matrix1, matrix2, raw_m2


def _p3_on_device(matrix1, matrix2, binary, filter_cols, target_cols, alpha=1.0) -> bool:
    """Can the whole preprocessing of p3alpha / rp3beta run inside the kernel call (SP_FLAG_P3_PREP, include/sp_knn.h)?
    Only for the plain call: matrix2 = matrix1.T, float32 data (float64 input is normalised in float64 by the reference),
    no `binary` (which throws the normalised values away), alpha > 0.

    Where the device form differs from similarity.py:410-415 / 477-483 (L1-normalise, `data ** alpha` on EVERY stored entry,
    only then eliminate_zeros inside s_plus): stored zeros are dropped BEFORE the power.  For alpha > 0 that is the same
    matrix (0 / norm = 0, 0 ** alpha = 0, the norm does not see zeros); for alpha <= 0 it is not (0 ** alpha is 1 or inf),
    so those calls take the host statement below.  An entry that underflows to 0 in the divide or the power would stay a stored
    (zero-valued) candidate on the device where the reference drops it: the library counts such entries and the call is then redone
    with the host statement (_abi.P3UnderflowError, see _run_p3; tests/golden/quirks_golden.npz pins the reference's result)."""
    from scipy.sparse import issparse
    if matrix2 is not None or binary or not issparse(matrix1) or matrix1.data.dtype != np.float32:
        return False
    if not (alpha > 0):
        return False
    if _host.multi_gpu_route() is not None:      # (the workers of the multi-GPU route get preprocessed matrices)
        return False
    return True      # (array-style column selectors too: the library drops those columns from the normalised m2, sp_knn_args.col_keep)


def _run_p3(matrix1, alpha, beta, k, shrink, shrink_type, threshold, target_rows, target_cols, filter_cols, verbose, format_output):
    stab, bayes, add = __get_shrink_values__(shrink, shrink_type)
    return _host._s_plus_impl(
        matrix1, None, 'none', 'none', 0.0, 0.0 if beta is None else beta, 1.0, 0.0, 0.0, 0.0 if beta is None else 1.0,
        1.0, 1.0, 0.5, 0.5, k, stab, bayes, add, threshold, False, target_rows, filter_cols, target_cols, verbose, format_output,
        p3_alpha=alpha, p3_depop_beta=beta)


def p3alpha(matrix1: sparray, matrix2: Optional[sparray] = None, alpha: float = 1.0, k: int = 100,
            shrink: float = 0.0, shrink_type: _Shrink = 'stabilized', threshold: float = 0.0,
            binary: bool = False, target_rows: _Rows = None, target_cols: _Cols = None,
            filter_cols: _Cols = None, verbose: bool = True, format_output: _Fmt = 'coo',
            num_threads: int = 0, block_size: Optional[int] = 0) -> sparray:
    """Top-k P3alpha: product of the two row-stochastic transition matrices, entries ^alpha."""
    if _p3_on_device(matrix1, matrix2, binary, filter_cols, target_cols, alpha):
        try:
            return _run_p3(matrix1, alpha, None, k, shrink, shrink_type, threshold, target_rows, target_cols, filter_cols, verbose, format_output)
        except _abi.P3UnderflowError:
            pass        # entries underflowed to 0.0: the reference drops them (s_plus.pyx:210-211) — the host statement below does too
    matrix1, matrix2, _ = _p3_inputs(matrix1, matrix2, alpha)
    return _run(matrix1, matrix2, {}, k, shrink, shrink_type, threshold, binary, target_rows, target_cols,
                filter_cols, verbose, format_output, num_threads, block_size)


def rp3beta(matrix1: sparray, matrix2: Optional[sparray] = None, alpha: float = 1.0, beta: float = 1.0,
            k: int = 100, shrink: float = 0.0, shrink_type: _Shrink = 'stabilized', threshold: float = 0.0,
            binary: bool = False, target_rows: _Rows = None, target_cols: _Cols = None,
            filter_cols: _Cols = None, verbose: bool = True, format_output: _Fmt = 'coo',
            num_threads: int = 0, block_size: Optional[int] = 0) -> sparray:
    """Top-k RP3beta: P3alpha divided by (column popularity of the raw matrix2)^beta."""
    if _p3_on_device(matrix1, matrix2, binary, filter_cols, target_cols, alpha):
        try:
            return _run_p3(matrix1, alpha, beta, k, shrink, shrink_type, threshold, target_rows, target_cols, filter_cols, verbose, format_output)
        except _abi.P3UnderflowError:
            pass        # (as in p3alpha)
    if matrix2 is None:
        matrix2 = matrix1.T
    pop_m2 = np.asarray(matrix2.sum(axis=0)).ravel()          # similarity.py:479 — BEFORE normalisation
    matrix1, matrix2, _ = _p3_inputs(matrix1, matrix2, alpha)
    return _run(matrix1, matrix2, dict(weight_depop_matrix2=pop_m2, p2=beta, l3=1), k, shrink, shrink_type,
                threshold, binary, target_rows, target_cols, filter_cols, verbose, format_output,
                num_threads, block_size)


def s_plus(matrix1: sparray, matrix2: Optional[sparray] = None, l1: float = 0.5, l2: float = 0.5,
           l3: float = 0.0, t1: float = 1.0, t2: float = 1.0, c1: float = 0.5, c2: float = 0.5,
           pop1: Optional[Union[str, np.ndarray]] = 'none', pop2: Optional[Union[str, np.ndarray]] = 'none',
           alpha: float = 1.0, beta1: float = 0.0, beta2: float = 0.0, k: int = 100, shrink: float = 0.0,
           shrink_type: _Shrink = 'stabilized', threshold: float = 0.0, binary: bool = False,
           target_rows: _Rows = None, target_cols: _Cols = None, filter_cols: _Cols = None,
           verbose: bool = True, format_output: _Fmt = 'coo', num_threads: int = 0,
           block_size: Optional[int] = 0) -> sparray:
    """Tversky + cosine + depopularisation hybrid:
    xy^alpha / (l1*tversky_den + l2*cosine_den + l3*pop1^beta1*pop2^beta2 + shrink)."""
    kw = dict(l1=l1, l2=l2, l3=l3, t1=t1, t2=t2, c1=c1, c2=c2, a1=alpha,
              weight_depop_matrix1=pop1, weight_depop_matrix2=pop2, p1=beta1, p2=beta2)
    return _run(matrix1, matrix2, kw, k, shrink, shrink_type, threshold, binary, target_rows, target_cols,
                filter_cols, verbose, format_output, num_threads, block_size)
