"""Host-side orchestration of the top-k similarity call.

Mirror of the reference's Cython orchestrator and its NumPy-level helpers:

    similaripy/cython_code/s_plus.pyx:95-433        def s_plus(...)            -> s_plus()
    similaripy/cython_code/s_plus_utils.pyx:19-125  validate_s_plus_inputs     -> validate_inputs()
    similaripy/cython_code/s_plus_utils.pyx:128-166 csr_sum                    -> csr_sum()
    similaripy/cython_code/s_plus_utils.pyx:169-278 norm builders              -> build_*()
    similaripy/cython_code/s_plus_utils.pyx:311-490 selectors / column filter  -> build_column_selector() ...
    similaripy/cython_code/utils.pyx:43-173         COO / CSR assembly         -> build_coo() / build_csr()

The work is split in three so tests can drive the kernel boundary directly:

    prepare(...)  -> KernelCall   every array/scalar compute_similarities_parallel receives
    run_hip(call) -> rows, cols, values, counts   (ctypes -> libsimilaripy_hip.so, HIP only)
    finish(...)   -> scipy.sparse output

Not reproduced on purpose: `_reorder_columns_by_popularity` (s_plus_utils.pyx:493-618) — a CPU
cache optimisation that only permutes slot order; `block_size` / `num_threads`
are accepted and ignored (results do not depend on them — tests/test_similarity.py:505).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import scipy.sparse as sp

from . import _abi

MODE_NONE, MODE_ARRAY, MODE_MATRIX = _abi.SP_SEL_NONE, _abi.SP_SEL_ARRAY, _abi.SP_SEL_MATRIX

_EMPTY_F32 = np.zeros(0, dtype=np.float32)
_EMPTY_I32 = np.zeros(0, dtype=np.int32)


# --------------------------------------------------------------------------------------------
# the kernel boundary
# --------------------------------------------------------------------------------------------
@dataclass
class KernelCall:
    """Arguments of compute_similarities_parallel<int,float> (s_plus.h:265-303) + the sizes a
    device copy needs.  All arrays are C-contiguous float32 / int32."""

    targets: np.ndarray
    m1_data: np.ndarray
    m1_indices: np.ndarray
    m1_indptr: np.ndarray
    m2_data: np.ndarray
    m2_indices: np.ndarray
    m2_indptr: np.ndarray
    n_rows_m1: int
    n_rows_m2: int
    n_output_cols: int
    k: int
    Xtversky: np.ndarray = field(default_factory=lambda: _EMPTY_F32)
    Ytversky: np.ndarray = field(default_factory=lambda: _EMPTY_F32)
    Xcosine: np.ndarray = field(default_factory=lambda: _EMPTY_F32)
    Ycosine: np.ndarray = field(default_factory=lambda: _EMPTY_F32)
    Xdepop: np.ndarray = field(default_factory=lambda: _EMPTY_F32)
    Ydepop: np.ndarray = field(default_factory=lambda: _EMPTY_F32)
    a1: float = 1.0
    l1: float = 0.0
    l2: float = 0.0
    l3: float = 0.0
    t1: float = 1.0
    t2: float = 1.0
    stabilized_shrink: float = 0.0
    bayesian_shrink: float = 0.0
    threshold: float = 0.0
    filter_mode: int = MODE_NONE
    filter_m_indptr: np.ndarray = field(default_factory=lambda: _EMPTY_I32)
    filter_m_indices: np.ndarray = field(default_factory=lambda: _EMPTY_I32)
    target_col_mode: int = MODE_NONE
    target_col_m_indptr: np.ndarray = field(default_factory=lambda: _EMPTY_I32)
    target_col_m_indices: np.ndarray = field(default_factory=lambda: _EMPTY_I32)
    m2_is_m1t: bool = False    # m2 = m1^T is built on the device (SP_FLAG_M2_IS_M1_T): the m2_* arrays are empty
    p3_alpha: Optional[float] = None         # SP_FLAG_P3_PREP: m1 / m2 = m1^T rows are L1-normalised and raised to this power on the device
    depop_rowsum_p2: Optional[float] = None  # SP_FLAG_DEPOP_ROWSUM: Ydepop = (row sums of the raw m1)^p2, built on the device
    m1_is_m2t: bool = False    # m1 = m2^T is built on the device (SP_FLAG_M1_IS_M2_T, matrix1 came as CSC): the m1_* arrays are empty
    norms_on_device: Optional[tuple] = None  # SP_FLAG_NORMS_ON_DEVICE: (c1, c2, additive_shrink); X/Y tversky / cosine vectors are empty
    check_m2_sorted: bool = False            # SP_FLAG_CHECK_SORTED: nobody has looked at the order inside the rows of the explicit m2 yet
    binary_on_device: bool = False           # SP_FLAG_BINARY: the value arrays are the caller's; the library writes ones into its uploaded copies
    col_keep: Optional[np.ndarray] = None    # with m2_is_m1t: uint8 [n_rows_m1], 0 = the output column is dropped while m2 is built (ARRAY selectors)

    @property
    def n_targets(self) -> int:
        return int(self.targets.shape[0])


# --------------------------------------------------------------------------------------------
# validation (s_plus_utils.pyx:19-125) — same checks, same exception types
# --------------------------------------------------------------------------------------------
def validate_inputs(matrix1, matrix2, weight_depop_matrix1, weight_depop_matrix2, k,
                    target_rows, filter_cols, target_cols, verbose, format_output) -> None:
    if not sp.issparse(matrix1):
        raise TypeError('matrix1 must be a sparse matrix')
    if not sp.issparse(matrix2):
        raise TypeError('matrix2 must be a sparse matrix')
    if matrix1.shape[1] != matrix2.shape[0]:
        raise ValueError(
            f'Incompatible matrix shapes: matrix1.shape[1]={matrix1.shape[1]} '
            f'must equal matrix2.shape[0]={matrix2.shape[0]}')
    if k < 1:
        raise ValueError(f'k must be >= 1, got {k}')

    def _check_weights(w, n, name):
        ok = False
        if isinstance(w, str):
            # the reference evaluates len('none') first; only the outcome is kept
            ok = w in ('none', 'sum') or len(w) == n
        else:
            ok = len(w) == n
        if not ok:
            raise ValueError(f'{name} must be array of length {n} or one of ("none", "sum"), got length {len(w)}')

    _check_weights(weight_depop_matrix1, matrix1.shape[0], 'weight_depop_matrix1')
    _check_weights(weight_depop_matrix2, matrix2.shape[1], 'weight_depop_matrix2')

    if target_rows is not None and len(target_rows) > matrix1.shape[0]:
        raise ValueError(
            f'target_rows length ({len(target_rows)}) cannot exceed matrix1.shape[0] ({matrix1.shape[0]})')

    expected = (matrix1.shape[0], matrix2.shape[1])
    for name, sel in (('filter_cols', filter_cols), ('target_cols', target_cols)):
        if sel is None:
            continue
        if not (sp.issparse(sel) or isinstance(sel, (list, np.ndarray))):
            raise TypeError(f'{name} must be a sparse matrix, list, numpy array, or None')
        if sp.issparse(sel) and sel.data.shape[0] != 0 and sel.shape != expected:
            raise ValueError(f'{name} shape {sel.shape} does not match expected shape {expected}')

    if not isinstance(verbose, bool):
        raise TypeError(f'verbose must be boolean, got {type(verbose).__name__}')
    if format_output not in ('coo', 'csr'):
        raise ValueError(f"format_output must be 'coo' or 'csr', got '{format_output}'")


# --------------------------------------------------------------------------------------------
# NumPy-level preprocessing (s_plus_utils.pyx:128-490)
# --------------------------------------------------------------------------------------------
def csr_sum(data: np.ndarray, indices: np.ndarray, indptr: np.ndarray, n_cols: int, axis: int) -> np.ndarray:
    """Row sums (axis=1: float32 np.add.reduceat, empty rows forced to 0) or column sums
    (axis=0: float64 np.bincount cast to float32) of a float32 CSR — s_plus_utils.pyx:128-166."""
    if axis == 1:
        n_rows = indptr.shape[0] - 1
        out = np.zeros(n_rows, dtype=np.float32)
        if data.shape[0] == 0:
            return out
        # reduce over the non-empty rows only: their starts are strictly increasing and < nnz, so
        # every segment is exactly one row (the reference reduces over all starts and then zeroes
        # the empty rows, :155-158 — same values; trailing empty rows would make that call raise)
        nonempty = np.diff(indptr) > 0
        out[nonempty] = np.add.reduceat(data, indptr[:-1][nonempty])
        return out
    if axis == 0:
        return np.bincount(indices, weights=data, minlength=n_cols).astype(np.float32, copy=False)
    raise ValueError(f"axis must be 0 or 1, got {axis}")


def build_squared_norms(m1_data, m1_indices, m1_indptr, n_cols_m1, m2_data, m2_indices, m2_indptr, n_cols_m2):
    """(sum x^2 per m1 row, sum y^2 per m2 column) — s_plus_utils.pyx:169-201."""
    sq1 = np.square(m1_data, dtype=np.float32)
    sq2 = np.square(m2_data, dtype=np.float32)
    return (csr_sum(sq1, m1_indices, m1_indptr, n_cols_m1, axis=1),
            csr_sum(sq2, m2_indices, m2_indptr, n_cols_m2, axis=0))


def squared_norms_m1t_hip(m1_data, m1_indptr, device: Optional[int] = None):
    """build_squared_norms_m1t on the GPU (sp_csr_row_sqsums_f32, include/sp_prep.h): the same two float32 vectors,
    bit for bit (NumPy's pairwise order for the row sums, float64 storage order for the column sums of m1^T).
    No CPU fallback: raises without a device."""
    _abi.require_device()
    n_rows = int(m1_indptr.shape[0]) - 1
    sq1 = np.empty(n_rows, dtype=np.float32)
    sq2 = np.empty(n_rows, dtype=np.float32)
    a = _abi.SpCsrSqsumsArgs()
    a.on_device = 0
    a.device = selected_device() if device is None else int(device)
    a.n_rows, a.nnz = n_rows, int(m1_data.shape[0])
    data, indptr = _abi.as_f32(m1_data), _abi.as_i32(m1_indptr)
    a.data = data.ctypes.data if data.size else None
    a.indptr = indptr.ctypes.data
    a.out_rows, a.out_cols_of_t = sq1.ctypes.data, sq2.ctypes.data
    if n_rows:
        _abi.call_row_sqsums(a)
    return sq1, sq2


def col_sums_hip(data, indices, n_cols: int, square: bool, device: Optional[int] = None) -> np.ndarray:
    """csr_sum(axis=0) of a float32 CSR on the GPU (sp_csr_col_sums_f32, include/sp_prep.h): np.bincount's float64
    accumulation, rounded to float32 (s_plus_utils.pyx:160-164); square: of data^2 (np.square in float32 first).  No CPU fallback."""
    _abi.require_device()
    out = np.zeros(n_cols, dtype=np.float32)
    a = _abi.SpCsrColsumsArgs()
    a.on_device = 0
    a.device = selected_device() if device is None else int(device)
    a.n_cols, a.square, a.nnz = int(n_cols), 1 if square else 0, int(data.shape[0])
    d, i = _abi.as_f32(data), _abi.as_i32(indices)
    a.data = d.ctypes.data if d.size else None
    a.indices = i.ctypes.data if i.size else None
    a.out = out.ctypes.data if n_cols else None
    if n_cols:
        _abi.call_col_sums(a)
    return out


def squared_norms_hip(m1_data, m1_indptr, m2_data, m2_indices, n_cols_m2: int):
    """build_squared_norms for an EXPLICIT matrix2 on the GPU: the row sums of m1^2 (sp_csr_row_sqsums_f32: NumPy's pairwise
    order, bit for bit) and the column sums of m2^2 (sp_csr_col_sums_f32: np.bincount's float64 accumulator)."""
    _abi.require_device()
    n_rows = int(m1_indptr.shape[0]) - 1
    sq1 = np.zeros(n_rows, dtype=np.float32)
    if n_rows and m1_data.shape[0]:
        a = _abi.SpCsrSqsumsArgs()
        a.on_device = 0
        a.device = selected_device()
        a.n_rows, a.nnz = n_rows, int(m1_data.shape[0])
        data, indptr = _abi.as_f32(m1_data), _abi.as_i32(m1_indptr)
        a.data, a.indptr = data.ctypes.data, indptr.ctypes.data
        a.out_rows, a.out_cols_of_t = sq1.ctypes.data, None
        _abi.call_row_sqsums(a)
    return sq1, col_sums_hip(m2_data, m2_indices, n_cols_m2, square=True)


def build_squared_norms_m1t(m1_data, m1_indptr):
    """build_squared_norms for m2 = m1^T without m2: the column sums of m2^2 are the row sums of m1^2, added up
    the way the reference adds the columns of m2 (np.bincount: float64, storage order — s_plus_utils.pyx:160-164;
    the storage order of a column of m1^T is the order of the row of m1)."""
    sq = np.square(m1_data, dtype=np.float32)
    n_rows = m1_indptr.shape[0] - 1
    sq1 = csr_sum(sq, None, m1_indptr, 0, axis=1)
    row_of = np.repeat(np.arange(n_rows, dtype=np.int32), np.diff(m1_indptr))
    sq2 = np.bincount(row_of, weights=sq, minlength=n_rows).astype(np.float32, copy=False)
    return sq1, sq2


def build_cosine_normalization(m1_sq, m2_sq, c1, c2, additive_shrink):
    """X=(sq1+add)^c1, Y=(sq2+add)^c2 in float32 — s_plus_utils.pyx:204-228."""
    c1, c2, add = float(np.float32(c1)), float(np.float32(c2)), float(np.float32(additive_shrink))
    return (np.power(m1_sq + add, c1, dtype=np.float32),
            np.power(m2_sq + add, c2, dtype=np.float32))


def build_depop_normalization(m1, m2, n_rows_m1, n_cols_m2, weight_spec1, weight_spec2, p1, p2, sums_on_device: bool = False,
                              m2_is_m1t: bool = False):
    """w^p, 'none' -> ones, 'sum' -> csr_sum^p — s_plus_utils.pyx:231-278.
    m1/m2 are (data, indices, indptr, n_cols) tuples of the float32 (or binarised) matrices.
    m2_is_m1t: m2 = m1^T is never built on the host (m2 is ignored): its column sums are taken from the rows of m1 with the
    arithmetic the reference applies to the columns of m2 (np.bincount: float64 running sum in storage order — the storage
    order of a column of m1^T is the order of the row of m1)."""
    p1, p2 = float(np.float32(p1)), float(np.float32(p2))

    def power(w, p):
        # a negative weight under a fractional exponent is NaN here as in the reference (s_plus_utils.pyx:257-276: np.power on
        # the same float32 values); NumPy's "invalid value" warning about it is the reference's too and says nothing new
        with np.errstate(invalid="ignore"):
            return np.power(w, p, dtype=np.float32)

    def one(spec, p, which):
        if isinstance(spec, (list, np.ndarray)):
            return power(spec, p)
        if spec == 'none':
            return np.ones(n_rows_m1 if which == 1 else n_cols_m2, dtype=np.float32)
        if spec == 'sum':
            if which == 2 and m2_is_m1t:
                d, _, ptr, _ = m1
                row_of = np.repeat(np.arange(n_rows_m1, dtype=np.int32), np.diff(ptr))
                if sums_on_device:      # (np.bincount's float64 pass over 64 M weights took 230 ms of such a call at the C2 size)
                    return power(col_sums_hip(d, row_of, n_rows_m1, square=False), p)
                return power(np.bincount(row_of, weights=d, minlength=n_rows_m1).astype(np.float32, copy=False), p)
            d, i, ptr, nc = m1 if which == 1 else m2
            if which == 2 and sums_on_device:      # column sums of m2: the device's np.bincount (sp_csr_col_sums_f32)
                return power(col_sums_hip(d, i, nc, square=False), p)
            return power(csr_sum(d, i, ptr, nc, axis=1 if which == 1 else 0), p)
        raise ValueError(f"Invalid weight_spec{which}: {spec}")

    return one(weight_spec1, p1, 1), one(weight_spec2, p2, 2)


def has_stored_zeros(data) -> bool:
    """Does a CSR value array hold explicit zeros (the question eliminate_zeros answers entry by entry)?  float32 zeros are
    the bit patterns 0 and 0x80000000: a minimum over the unsigned view and one over the signed view find them at memory speed
    (np.count_nonzero on float32 takes 4x as long: 120 ms at 64 M entries)."""
    if data.shape[0] == 0:
        return False
    if data.dtype == np.float32 and data.flags.c_contiguous:
        return bool(data.view(np.uint32).min() == 0) or bool(data.view(np.int32).min() == np.iinfo(np.int32).min)
    return bool(np.count_nonzero(data) != data.shape[0])


def build_column_selector(cols, verify_order: bool = True):
    """(mode, indptr, indices): sparse with data -> MATRIX (CSR, zeros removed, sorted rows);
    non-empty list/array -> ARRAY; else NONE — s_plus_utils.pyx:311-361.
    verify_order=False: scipy's cached has_sorted_indices flag is taken at its word here because the library looks at the order itself
    where the selector goes anyway (run_host: sp_rows_sorted_kernel on the uploaded rows -> UnsortedRowsError -> the caller comes back
    with verify_order=True); True: one vectorised pass over the indices here (60 ms at 64 M entries)."""
    if sp.issparse(cols) and cols.data.shape[0] != 0:
        m = cols.tocsr()
        # the caller's matrix is never modified; it is copied only when something has to change (an URM used as its own
        # filter — the common case — is canonical already, and its copy was most of this call's host time: 64 M entries)
        # (has_sorted_indices is a CACHED scipy flag that in-place edits of .indices do not refresh: a stale True would hand unsorted rows
        # to the kernels' binary search over the selector row — the order is looked at, _rows_sorted, one vectorised pass)
        in_order = m.has_sorted_indices if not verify_order else _rows_sorted(np.asarray(m.indices), np.asarray(m.indptr))
        dirty = has_stored_zeros(m.data) or not in_order
        if dirty:
            if m is cols:
                m = m.copy()
            m.eliminate_zeros()
            m.has_sorted_indices = False      # (never trust the cached flag on the way to a sort)
            m.sort_indices()
        return MODE_MATRIX, np.ascontiguousarray(m.indptr, dtype=np.int32), np.ascontiguousarray(m.indices, dtype=np.int32)
    if isinstance(cols, (list, np.ndarray)) and len(cols) != 0:
        return MODE_ARRAY, _EMPTY_I32, _EMPTY_I32
    return MODE_NONE, _EMPTY_I32, _EMPTY_I32


def compute_target_columns(filter_cols, target_cols, n_cols: int) -> np.ndarray:
    """Columns that survive array-style target/filter lists — s_plus_utils.pyx:364-421.
    Out-of-range ids are silently dropped (:411,:418)."""
    def is_empty(c):
        return c is None or (isinstance(c, (list, np.ndarray)) and len(c) == 0)

    def is_matrix(c):
        return sp.issparse(c) and c.data.shape[0] != 0

    f_empty, t_empty = is_empty(filter_cols), is_empty(target_cols)
    f_mat, t_mat = is_matrix(filter_cols), is_matrix(target_cols)
    if (f_empty and t_empty) or (f_mat and t_mat) or (f_mat and t_empty) or (t_mat and f_empty):
        return np.arange(n_cols, dtype=np.int32)
    if not t_empty and not t_mat:
        mask = np.zeros(n_cols, dtype=bool)
        idx = np.asarray(target_cols, dtype=np.int32)
        mask[idx[(idx >= 0) & (idx < n_cols)]] = True
    else:
        mask = np.ones(n_cols, dtype=bool)
    if not f_empty and not f_mat:
        idx = np.asarray(filter_cols, dtype=np.int32)
        mask[idx[(idx >= 0) & (idx < n_cols)]] = False
    return np.flatnonzero(mask).astype(np.int32, copy=False)


def column_keep_mask(filter_cols, target_cols, n_cols: int) -> np.ndarray:
    """The same selection as a byte mask (sp_knn_args.col_keep) — without the detour through the list of kept ids."""
    def arr(c):
        return isinstance(c, (list, np.ndarray)) and len(c) != 0

    if not arr(filter_cols) and not arr(target_cols):
        keep = np.zeros(n_cols, dtype=np.uint8)
        keep[compute_target_columns(filter_cols, target_cols, n_cols)] = 1
        return keep
    if arr(target_cols):
        keep = np.zeros(n_cols, dtype=np.uint8)
        idx = np.asarray(target_cols, dtype=np.int32)
        keep[idx[(idx >= 0) & (idx < n_cols)]] = 1
    else:
        keep = np.ones(n_cols, dtype=np.uint8)
    if arr(filter_cols):
        idx = np.asarray(filter_cols, dtype=np.int32)
        keep[idx[(idx >= 0) & (idx < n_cols)]] = 0
    return keep


def filter_matrix_columns(data, indices, indptr, n_cols: int, keep_cols: np.ndarray):
    """Drop CSR entries whose column is not in keep_cols; column ids are preserved
    (s_plus_utils.pyx:424-490, the two typed loops become one mask + cumulative count)."""
    mask = np.zeros(n_cols, dtype=bool)
    kc = np.asarray(keep_cols, dtype=np.int32)
    mask[kc[(kc >= 0) & (kc < n_cols)]] = True
    keep = mask[indices]
    csum = np.concatenate(([0], np.cumsum(keep, dtype=np.int64)))
    new_indptr = csum[indptr].astype(np.int32)
    return (np.ascontiguousarray(data[keep], dtype=np.float32),
            np.ascontiguousarray(indices[keep], dtype=np.int32),
            new_indptr)


# --------------------------------------------------------------------------------------------
# output assembly (utils.pyx:43-173, coo_to_csr.h:28-71)
# --------------------------------------------------------------------------------------------
def build_coo(rows, cols, values, n_rows: int, n_cols: int) -> sp.coo_array:
    """COO over the raw slot arrays, zero padding included (utils.pyx:43-64).
    The arrays come from the kernel (indices in range by construction): they are attached to an empty matrix instead of
    going through the constructor, whose validation takes the minimum and maximum of both index arrays (50 ms at 10^8
    entries).  Any scipy without these attributes gets the constructor."""
    if (rows.dtype == np.int32 and cols.dtype == np.int32 and values.dtype == np.float32 and rows.ndim == 1
            and rows.shape == cols.shape == values.shape and max(n_rows, n_cols) <= np.iinfo(np.int32).max):
        try:
            res = sp.coo_array((n_rows, n_cols), dtype=np.float32)
            if not hasattr(res, "coords"):
                raise AttributeError("coords")
            res.coords = (rows, cols)
            res.data = values
            res.has_canonical_format = False
            if res.nnz == values.shape[0] and res.shape == (n_rows, n_cols):
                return res
        except Exception:       # noqa: BLE001  (an older / newer scipy: the documented way)
            pass
    return sp.coo_array((values, (rows, cols)), shape=(n_rows, n_cols), dtype=np.float32)


def build_csr(targets, cols, values, counts, k: int, n_rows: int, n_cols: int) -> sp.csr_array:
    """CSR with padding and genuine zeros removed, rows not in `targets` empty, entries of a
    row in slot order (utils.pyx:141-173 -> coo_to_csr.h:28-71 -> eliminate_zeros, s_plus.pyx:424).
    The reference counting-sorts n_targets*k triples incl. padding; the per-slot counts the
    kernel returns let us skip the padding up front — same result."""
    n_targets = targets.shape[0]
    total = n_targets * k
    idx_dtype = np.int32 if max(total, n_cols) <= np.iinfo(np.int32).max else np.int64
    counts = counts.astype(np.int64, copy=False)
    if n_targets == 0:
        return sp.csr_array((n_rows, n_cols), dtype=np.float32)
    if bool((counts == k).all()):
        v, c = values, cols                      # every slot full (the usual case at scale): nothing to strip, no copies
    else:
        valid = (np.arange(k, dtype=np.int64)[None, :] < counts[:, None]).ravel()
        v = values[valid]
        c = cols[valid]
    row_nnz = np.zeros(n_rows, dtype=np.int64)
    strictly_increasing = n_targets == 1 or bool(np.all(targets[1:] > targets[:-1]))
    if strictly_increasing:
        row_nnz[targets] = counts
    else:
        # general (unsorted / repeated target_rows): stable counting sort by row id
        r = np.repeat(targets.astype(np.int64), counts)
        order = np.argsort(r, kind='stable')
        v, c = v[order], c[order]
        np.add.at(row_nnz, targets, counts)
    indptr = np.zeros(n_rows + 1, dtype=idx_dtype)
    np.cumsum(row_nnz, out=indptr[1:])
    res = sp.csr_array((v, c.astype(idx_dtype, copy=False), indptr), shape=(n_rows, n_cols), dtype=np.float32)
    if has_stored_zeros(res.data):      # (the in-place pass of eliminate_zeros costs more than the check)
        res.eliminate_zeros()
    return res


# --------------------------------------------------------------------------------------------
# prepare / run / finish
# --------------------------------------------------------------------------------------------
def _say(verbose: bool, msg: str) -> None:
    # the reference drives a C++ progress bar on stderr (s_plus.pyx:199-202, progress_bar.h:199-208); a per-row host callback has no
    # device analogue: the phase names are reported here, and the library prints "rows done a / n" whenever a chunk of result rows has
    # reached the host (SP_FLAG_PROGRESS: four chunks for a large call, one line for a small one)
    if verbose:
        print(f"[similaripy_amd] {msg}", file=sys.stderr, flush=True)


def _csr_f32_i32(m, binary: bool, check_zeros: bool = True):
    """CSR view with zeros eliminated, float32 data (ones if `binary`), int32 indices/indptr.
    Unlike the reference (s_plus.pyx:210-211) the caller's matrix is never modified.
    check_zeros=False: the caller leaves the search for stored zeros to the device (SP_FLAG_CHECK_ZEROS)."""
    m = m.tocsr()
    if check_zeros and m.data.shape[0] and has_stored_zeros(m.data):
        m = m.copy()
        m.eliminate_zeros()
    if m.nnz > np.iinfo(np.int32).max:
        raise ValueError("matrix has more than 2^31-1 stored entries (int32 index limit, s_plus.pyx:241-244)")
    if binary:
        data = np.ones(m.data.shape[0], dtype=np.float32)
    else:
        data = np.ascontiguousarray(m.data, dtype=np.float32)
    return m, data, np.ascontiguousarray(m.indices, dtype=np.int32), np.ascontiguousarray(m.indptr, dtype=np.int32)


def prepare(matrix1, matrix2=None, weight_depop_matrix1='none', weight_depop_matrix2='none',
            p1=0.0, p2=0.0, a1=1.0, l1=0.0, l2=0.0, l3=0.0, t1=1.0, t2=1.0, c1=0.5, c2=0.5, k=100,
            stabilized_shrink=0.0, bayesian_shrink=0.0, additive_shrink=0.0, threshold=0.0,
            binary=False, target_rows=None, filter_cols=None, target_cols=None,
            verbose=False, format_output='csr', m2_on_device=False, check_zeros=True,
            p3_alpha=None, p3_depop_beta=None, norms_on_device=False, csc_direct=False, keep_on_device=False,
            binary_on_device=False, m2_sorted_on_device=False, selectors_sorted_on_device=False) -> KernelCall:
    """Everything s_plus.pyx does before the `with nogil:` block (:168-353).

    check_zeros=False: stored zeros are looked for on the device (run_hip(check_zeros=True)) instead of here.
    p3_alpha / p3_depop_beta: the call is p3alpha / rp3beta on the RAW matrix1 (matrix2=None): normalisation, power and
    the column popularity are left to the device (SP_FLAG_P3_PREP / SP_FLAG_DEPOP_ROWSUM); needs m2_on_device.

    m2_on_device: for the `matrix2=None` call, leave the transpose (s_plus.pyx:169-170, 205-206) to the device
    (SP_FLAG_M2_IS_M1_T, include/sp_prep.h): m2 is never built on the host, its column norms are taken from the
    rows of m1 with the arithmetic the reference applies to the columns of m2.  ARRAY column selectors become a mask the
    device applies while it builds m2 (KernelCall.col_keep); depopularisation weights ('none' / 'sum' / arrays) are taken
    from m1; with p3_alpha the mask is applied after the rows of m2 were normalised.

    norms_on_device: with the device-side transpose, leave _build_squared_norms / _build_cosine_normalization to the same
    library call (SP_FLAG_NORMS_ON_DEVICE): the call carries (c1, c2, additive_shrink) instead of the vectors.
    csc_direct: a CSC matrix1 (`URM.T` of a CSR URM: the documented item-item call) is not converted on the host
    (matrix1.tocsr(), s_plus.pyx:205-206): its arrays ARE the CSR of matrix2 = matrix1.T, and m1 is built from them on
    the device (SP_FLAG_M1_IS_M2_T).  Needs norms_on_device (there is no m1 on the host to take norms from).
    keep_on_device: ARRAY selectors on an EXPLICIT matrix2 are left to the library's host-mode entry too (col_keep: the
    uploaded m2 is compacted on the device) instead of _filter_matrix_columns here; such a call cannot go to DeviceProblem."""
    m2_from_m1 = matrix2 is None
    if matrix2 is None:
        matrix2 = matrix1.T
    k = int(k)
    validate_inputs(matrix1, matrix2, weight_depop_matrix1, weight_depop_matrix2, k,
                    target_rows, filter_cols, target_cols, verbose, format_output)
    if k > matrix2.shape[1]:
        k = matrix2.shape[1]                                   # s_plus.pyx:187-188

    if target_rows is None:
        targets = np.arange(matrix1.shape[0], dtype=np.int32)
    else:
        targets = np.ascontiguousarray(np.asarray(target_rows, dtype=np.int32))
        if targets.size and (targets.min() < 0 or targets.max() >= matrix1.shape[0]):
            # the reference does not check (s_plus.pyx:191-196: out-of-range is UB there)
            raise ValueError("target_rows contains row ids outside matrix1")

    # (selectors_sorted_on_device: the host-mode entry checks the order inside the rows of a MATRIX selector on the device; a caller of
    # DeviceProblem / the oracle gets the host pass)
    sel_f = build_column_selector(filter_cols, verify_order=not selectors_sorted_on_device)
    sel_t = build_column_selector(target_cols, verify_order=not selectors_sorted_on_device)
    p3 = p3_alpha is not None
    # ARRAY selectors drop whole columns of m2 (s_plus_utils.pyx:364-490): with the device-side transpose that is a mask over
    # the rows of m1 it reads (KernelCall.col_keep).  With p3_alpha the library applies the mask to the NORMALISED m2 instead (the
    # reference normalises the rows of m2 first and drops the columns afterwards: masking earlier would change the row sums).
    arr_sel = sel_f[0] == MODE_ARRAY or sel_t[0] == MODE_ARRAY
    on_dev = bool(m2_on_device) and m2_from_m1 and (l3 == 0 or not p3 or p3_depop_beta is not None)
    if p3 and not on_dev:
        raise ValueError("p3_alpha needs the device-side transpose (matrix2=None)")
    # (norms_on_device with an explicit matrix2: host mode builds both vectors from its uploaded copies, SP_FLAG_NORMS_ON_DEVICE)
    m2_explicit_dev = bool(m2_on_device) and not m2_from_m1
    dev_norms = (on_dev or m2_explicit_dev) and bool(norms_on_device) and (l1 != 0 or l2 != 0)
    # binary=True (s_plus.pyx:214-217: data = ones after eliminate_zeros): when nothing on the host reads the values (device-built m2
    # and norms, no depop weights) they go up as they are and the library writes the ones into its copies (SP_FLAG_BINARY) — no array
    # of ones is built or uploaded, and the stored-zero check can stay on the device.  Otherwise the ones are made here, and then the
    # zero check has to happen here too (the device would only see ones).
    bin_dev = bool(binary) and bool(binary_on_device) and (on_dev or m2_explicit_dev) and not p3 and l3 == 0 and (dev_norms or (l1 == 0 and l2 == 0))
    host_binary = bool(binary) and not bin_dev
    check_zeros = bool(check_zeros) or host_binary
    # (a 'sum' weight of matrix1 is its ROW sums in the reference's float32 reduceat order, s_plus_utils.pyx:128-158: that needs the
    # CSR of matrix1 on the host, which the CSC route never builds)
    w1_rowsum = l3 != 0 and isinstance(weight_depop_matrix1, str) and weight_depop_matrix1 == 'sum'
    csc = (on_dev and bool(csc_direct) and not arr_sel and not w1_rowsum and getattr(matrix1, "format", None) == "csc" and (dev_norms or (l1 == 0 and l2 == 0))
           and matrix1.nnz <= np.iinfo(np.int32).max
           and not (check_zeros and matrix1.data.shape[0] and has_stored_zeros(matrix1.data)))
    if csc:
        # (data, indices, indptr) of the CSC matrix1 are the CSR arrays of matrix2 = matrix1.T (s_plus.pyx:169-170)
        n_rows_m1, n_rows_m2 = matrix1.shape
        n_output_cols = n_rows_m1
        m1_data, m1_indices, m1_indptr = _EMPTY_F32, _EMPTY_I32, _EMPTY_I32
        m2_data = np.ones(matrix1.data.shape[0], dtype=np.float32) if host_binary else np.ascontiguousarray(matrix1.data, dtype=np.float32)
        m2_indices = np.ascontiguousarray(matrix1.indices, dtype=np.int32)
        m2_indptr = np.ascontiguousarray(matrix1.indptr, dtype=np.int32)
    else:
        m1, m1_data, m1_indices, m1_indptr = _csr_f32_i32(matrix1, host_binary, check_zeros)
        n_rows_m1, n_rows_m2 = m1.shape
        if on_dev:
            m2_data, m2_indices, m2_indptr = _EMPTY_F32, _EMPTY_I32, _EMPTY_I32
            n_output_cols = n_rows_m1
        else:
            m2, m2_data, m2_indices, m2_indptr = _csr_f32_i32(matrix2, host_binary, check_zeros)
            n_output_cols = m2.shape[1]

    # all scalar parameters are C floats in the reference (s_plus.pyx:100-113)
    f32 = lambda x: float(np.float32(x))  # noqa: E731
    a1, l1, l2, l3, t1, t2 = map(f32, (a1, l1, l2, l3, t1, t2))
    stabilized_shrink, bayesian_shrink, threshold = map(f32, (stabilized_shrink, bayesian_shrink, threshold))

    call = KernelCall(
        targets=targets, m1_data=m1_data, m1_indices=m1_indices, m1_indptr=m1_indptr,
        m2_data=m2_data, m2_indices=m2_indices, m2_indptr=m2_indptr,
        n_rows_m1=n_rows_m1, n_rows_m2=n_rows_m2, n_output_cols=n_output_cols, k=k,
        a1=a1, l1=l1, l2=l2, l3=l3, t1=t1, t2=t2,
        stabilized_shrink=stabilized_shrink, bayesian_shrink=bayesian_shrink, threshold=threshold,
        m2_is_m1t=on_dev and not csc, m1_is_m2t=csc, binary_on_device=bin_dev)
    if p3:
        call.p3_alpha = f32(p3_alpha)

    if dev_norms:
        call.norms_on_device = (f32(c1), f32(c2), f32(additive_shrink))
    elif l1 != 0 or l2 != 0:
        if on_dev:
            sq1, sq2 = squared_norms_m1t_hip(m1_data, m1_indptr)
        elif m2_on_device:      # (the public call with an explicit matrix2: both norm vectors from the device)
            sq1, sq2 = squared_norms_hip(m1_data, m1_indptr, m2_data, m2_indices, n_output_cols)
        else:
            sq1, sq2 = build_squared_norms(m1_data, m1_indices, m1_indptr, n_rows_m2,
                                           m2_data, m2_indices, m2_indptr, n_output_cols)
        if l1 != 0:
            call.Xtversky, call.Ytversky = sq1, sq2
        if l2 != 0:
            call.Xcosine, call.Ycosine = build_cosine_normalization(sq1, sq2, c1, c2, additive_shrink)
    if l3 != 0 and p3 and p3_depop_beta is not None:
        # rp3beta: Xdepop = ones ('none'), Ydepop = popularity^beta from the raw rows of m1 on the device (s_plus_utils.pyx:257-276)
        call.Xdepop = np.ones(n_rows_m1, dtype=np.float32)
        call.depop_rowsum_p2 = f32(p3_depop_beta)
    elif l3 != 0:
        call.Xdepop, call.Ydepop = build_depop_normalization(
            (m1_data, m1_indices, m1_indptr, n_rows_m2), (m2_data, m2_indices, m2_indptr, n_output_cols),
            n_rows_m1, n_output_cols, weight_depop_matrix1, weight_depop_matrix2, p1, p2, sums_on_device=bool(m2_on_device),
            m2_is_m1t=on_dev and not csc)
        call.Xdepop = np.ascontiguousarray(call.Xdepop, dtype=np.float32)
        call.Ydepop = np.ascontiguousarray(call.Ydepop, dtype=np.float32)

    call.filter_mode, call.filter_m_indptr, call.filter_m_indices = sel_f
    call.target_col_mode, call.target_col_m_indptr, call.target_col_m_indices = sel_t
    if on_dev:
        if arr_sel:
            call.col_keep = column_keep_mask(filter_cols, target_cols, n_output_cols)
        return call        # (the device builds m2 with ascending column ids; SP_FLAG_M1_IS_M2_T checks those of the caller's)
    if arr_sel and keep_on_device:
        call.col_keep = column_keep_mask(filter_cols, target_cols, n_output_cols)
    elif arr_sel:
        keep = compute_target_columns(filter_cols, target_cols, n_output_cols)
        call.m2_data, call.m2_indices, call.m2_indptr = filter_matrix_columns(
            m2_data, m2_indices, m2_indptr, n_output_cols, keep)

    # The kernel windows the columns when they exceed its LDS tile and then needs ascending column
    # ids inside each m2 row — what the reference's blocked path gets from sort_indices()
    # (s_plus_utils.pyx:562).  Transposes and scipy-built CSR already are.
    # (m2_sorted_on_device: the library looks, where m2 goes anyway — SP_FLAG_CHECK_SORTED, UnsortedRowsError — instead of a pass here)
    call.check_m2_sorted = bool(m2_sorted_on_device)
    if not m2_sorted_on_device and not _rows_sorted(call.m2_indices, call.m2_indptr):
        tmp = sp.csr_array((call.m2_data.copy(), call.m2_indices.copy(), call.m2_indptr.copy()),
                           shape=(n_rows_m2, n_output_cols))
        tmp.sort_indices()
        call.m2_data = np.ascontiguousarray(tmp.data, dtype=np.float32)
        call.m2_indices = np.ascontiguousarray(tmp.indices, dtype=np.int32)
        call.m2_indptr = np.ascontiguousarray(tmp.indptr, dtype=np.int32)
    return call


def _rows_sorted(indices: np.ndarray, indptr: np.ndarray) -> bool:
    if indices.shape[0] < 2:
        return True
    bad = indices[1:] <= indices[:-1]          # also flags duplicates: harmless, just a re-sort
    if not bad.any():
        return True
    # a descent is only legal exactly at a row boundary
    starts = np.zeros(indices.shape[0], dtype=bool)
    s = indptr[1:-1]
    starts[s[s < indices.shape[0]]] = True
    return not np.any(bad & ~starts[1:])


def selected_device() -> int:
    return int(os.environ.get("SIMILARIPY_AMD_DEVICE", "0"))


def run_hip(call: KernelCall, device: Optional[int] = None, table_slots: int = 0, threads_per_wg: int = 0,
            num_wgs: int = 0, load_pct: int = 0, time_kernel: bool = False, static_sched: bool = False,
            no_sparse_path: bool = False, no_fold: bool = False, want_rows: bool = True,
            check_zeros: bool = False, csr_out: bool = False, dbg: int = 0, devices=None, progress: bool = False):
    """The `with nogil:` block of s_plus.pyx:359-384, on the GPU: host buffers in, host buffers out
    through the C ABI (include/sp_knn.h).  Returns rows, cols, values, counts[, info].

    check_zeros: SP_FLAG_CHECK_ZEROS — raises _abi.ExplicitZerosError when m1 / m2 hold stored zeros.
    csr_out: SP_FLAG_CSR_OUT — the CSR result is assembled on the device (any order of the targets, repeats included); returns
    (indptr, indices, data) of the final matrix instead (views of the buffers the library filled).
    progress: SP_FLAG_PROGRESS — "rows done a / n" on stderr whenever a chunk of result rows has reached the host (verbose=True).
    devices: a list of HIP ordinals — sp_knn_args.n_devices / device_ids (ABI 5): the library cuts the target list into
    cost-balanced contiguous slices and runs slice r on devices[r], one host thread per device, inside this ONE call
    (with csr_out the targets must ascend strictly)."""
    if devices is not None and len(list(devices)) == 0:
        raise ValueError("devices is empty")
    _abi.require_device()
    n, k = call.n_targets, call.k
    want_rows = want_rows and not csr_out
    rows = np.empty(n * k, dtype=np.int32) if want_rows else None      # (slot i's rows are all targets[i]: CSR assembly does not read them)
    # (CSR out with a MATRIX target selector: at most one entry per listed column — the library writes, and touches, no more than that;
    # 800 MB of slot arrays for 200 k entries cost 40 ms of page faults and unmapping at the C2 size)
    # — when the targets ascend strictly: a target that repeats emits its row once per repeat (target_rows=[7, 7, 7] against a
    # list of 5 columns in row 7 is 15 entries), so any other order keeps the full n * k, as include/sp_knn.h says)
    n_out = n * k
    if (csr_out and call.target_col_mode == MODE_MATRIX and devices is None      # (several devices write their pieces at slot offsets)
            and (n < 2 or bool(np.all(np.diff(np.asarray(call.targets, dtype=np.int64)) > 0)))):
        n_out = min(n_out, int(call.target_col_m_indices.shape[0]))
    cols = np.empty(n_out, dtype=np.int32)
    values = np.empty(n_out, dtype=np.float32)
    counts = np.empty(n, dtype=np.int32) if not csr_out else None
    csr_indptr = np.zeros(call.n_rows_m1 + 1, dtype=np.int32) if csr_out else None

    a = _abi.SpKnnArgs()
    a.flags = ((0 if want_rows else _abi.SP_FLAG_NO_ROWS_OUT) | (_abi.SP_FLAG_TIME_KERNEL | _abi.SP_FLAG_PHASE_TIMERS if time_kernel else 0) | (_abi.SP_FLAG_STATIC_SCHED if static_sched else 0)
               | (_abi.SP_FLAG_NO_SPARSE_PATH if no_sparse_path else 0) | (_abi.SP_FLAG_NO_FOLD if no_fold else 0)
               | (_abi.SP_FLAG_CHECK_ZEROS if check_zeros else 0) | (_abi.SP_FLAG_CSR_OUT if csr_out else 0)
               | (_abi.SP_FLAG_BINARY if call.binary_on_device else 0) | (_abi.SP_FLAG_CHECK_SORTED if call.check_m2_sorted else 0)
               | (_abi.SP_FLAG_PROGRESS if progress else 0))
    if call.p3_alpha is not None:
        a.flags |= _abi.SP_FLAG_P3_PREP
        a.p3_alpha = call.p3_alpha
    if call.depop_rowsum_p2 is not None:
        a.flags |= _abi.SP_FLAG_DEPOP_ROWSUM
        a.depop_p2 = call.depop_rowsum_p2
    a.on_device = 0
    a.device = selected_device() if device is None else int(device)
    a.n_targets, a.n_rows_m1, a.n_rows_m2, a.n_output_cols = n, call.n_rows_m1, call.n_rows_m2, call.n_output_cols
    a.nnz_m1, a.nnz_m2 = int(call.m1_data.shape[0]), int(call.m2_data.shape[0])
    if call.m2_is_m1t:
        a.flags |= _abi.SP_FLAG_M2_IS_M1_T
    if call.m1_is_m2t:
        a.flags |= _abi.SP_FLAG_M1_IS_M2_T
    if call.norms_on_device is not None:
        a.flags |= _abi.SP_FLAG_NORMS_ON_DEVICE
        a.norm_c1, a.norm_c2, a.norm_add = call.norms_on_device
    keep = []  # keep converted arrays alive across the call
    if devices is not None:
        dv = np.ascontiguousarray(np.asarray(list(devices), dtype=np.int32))
        keep.append(dv)
        a.n_devices, a.device_ids = int(dv.size), dv.ctypes.data
        a.device = int(dv[0])

    def f32(x):
        x = _abi.as_f32(x); keep.append(x); return x.ctypes.data if x.size else None

    def i32(x):
        x = _abi.as_i32(x); keep.append(x); return x.ctypes.data if x.size else None

    a.targets = i32(call.targets)
    if call.col_keep is not None:
        ck = np.ascontiguousarray(call.col_keep, dtype=np.uint8); keep.append(ck)
        a.col_keep = ck.ctypes.data if ck.size else None
    a.m1_data, a.m1_indices, a.m1_indptr = f32(call.m1_data), i32(call.m1_indices), i32(call.m1_indptr)
    a.m2_data, a.m2_indices, a.m2_indptr = f32(call.m2_data), i32(call.m2_indices), i32(call.m2_indptr)
    a.Xtversky, a.Ytversky = f32(call.Xtversky), f32(call.Ytversky)
    a.Xcosine, a.Ycosine = f32(call.Xcosine), f32(call.Ycosine)
    a.Xdepop, a.Ydepop = f32(call.Xdepop), f32(call.Ydepop)
    a.a1, a.l1, a.l2, a.l3, a.t1, a.t2 = call.a1, call.l1, call.l2, call.l3, call.t1, call.t2
    a.stabilized_shrink, a.bayesian_shrink, a.threshold = call.stabilized_shrink, call.bayesian_shrink, call.threshold
    a.k = k
    a.filter_mode = call.filter_mode
    a.filter_m_indptr, a.filter_m_indices = i32(call.filter_m_indptr), i32(call.filter_m_indices)
    a.filter_nnz = int(call.filter_m_indices.shape[0])
    a.target_col_mode = call.target_col_mode
    a.target_col_m_indptr, a.target_col_m_indices = i32(call.target_col_m_indptr), i32(call.target_col_m_indices)
    a.target_col_nnz = int(call.target_col_m_indices.shape[0])
    a.rows, a.cols, a.values = (rows.ctypes.data if want_rows else None), cols.ctypes.data, values.ctypes.data
    a.out_counts = counts.ctypes.data if counts is not None else None
    a.csr_indptr = csr_indptr.ctypes.data if csr_out else None
    a.table_slots, a.threads_per_wg, a.num_wgs, a.load_pct = table_slots, threads_per_wg, num_wgs, load_pct
    a.reserved[0] = int(dbg)          # ablation word of the library (tests / profiling only; 1024 = force the 64-bit-offset kernel variant)
    if n > 0:
        if os.environ.get("SIMILARIPY_AMD_TRACE", "0") not in ("", "0"):
            import time
            t0 = time.perf_counter()
            _abi.call_knn(a)
            print(f"[similaripy_amd] sp_knn_f32_i32 as seen from Python      {1e3 * (time.perf_counter() - t0):8.2f} ms", file=sys.stderr, flush=True)
        else:
            _abi.call_knn(a)
    if csr_out:
        nnz = int(a.csr_nnz) if n > 0 else 0
        return csr_indptr, cols[:nnz], values[:nnz]
    if time_kernel:
        return rows, cols, values, counts, {"kernel_ms": float(a.kernel_ms), "passes_total": int(a.passes_total), "phase_cycles": [int(x) for x in a.phase_cycles], "num_wgs": int(a.num_wgs_used),
                                             "sparse_kernel_ms": int(a.reserved[1]) / 1e3, "generic_kernel_ms": int(a.reserved[2]) / 1e3, "transpose_ms": int(a.reserved[3]) / 1e3}
    return rows, cols, values, counts


def slot_rows(targets: np.ndarray, counts: np.ndarray, k: int) -> np.ndarray:
    """The `rows` array of the kernel's slots without downloading it: every entry of slot i is in row targets[i], the
    padding behind counts[i] is (0, 0, 0.0) (s_plus.h:246-262 leaves the calloc'ed tail untouched)."""
    n = int(targets.shape[0])
    rows = np.repeat(np.asarray(targets, dtype=np.int32), k)
    if n and not bool((counts == k).all()):
        rows.reshape(n, k)[np.arange(k, dtype=np.int32)[None, :] >= counts[:, None]] = 0
    return rows


def finish(call: KernelCall, rows, cols, values, counts, format_output: str):
    """Everything after the kernel in s_plus.pyx:386-433.  rows=None: rebuilt from the targets and the per-slot counts."""
    if format_output == 'coo':
        if rows is None:
            rows = slot_rows(call.targets, counts, call.k)
        return build_coo(rows, cols, values, call.n_rows_m1, call.n_output_cols)
    return build_csr(call.targets, cols, values, counts, call.k, call.n_rows_m1, call.n_output_cols)


def s_plus(matrix1, matrix2=None, weight_depop_matrix1='none', weight_depop_matrix2='none',
           p1=0.0, p2=0.0, a1=1.0, l1=0.0, l2=0.0, l3=0.0, t1=1.0, t2=1.0, c1=0.5, c2=0.5, k=100,
           stabilized_shrink=0.0, bayesian_shrink=0.0, additive_shrink=0.0, threshold=0.0,
           binary=False, target_rows=None, filter_cols=None, target_cols=None,
           verbose=True, format_output='csr', num_threads=0, block_size=0):
    """Top-K similarity between the rows of two sparse matrices — same signature, defaults and
    result as ``similaripy.cython_code.s_plus.s_plus`` (s_plus.pyx:95-123), computed on the GPU.

    ``num_threads`` and ``block_size`` are CPU tuning knobs of the reference; they are accepted
    and ignored (they never change the result).
    """
    return _s_plus_impl(matrix1, matrix2, weight_depop_matrix1, weight_depop_matrix2, p1, p2, a1, l1, l2, l3,
                        t1, t2, c1, c2, k, stabilized_shrink, bayesian_shrink, additive_shrink, threshold,
                        binary, target_rows, filter_cols, target_cols, verbose, format_output)


_MULTI_GPU_ROUTE: list = []      # set by multi_gpu.similarity() around a wrapper call


def multi_gpu_route():
    """(devices, chunk_rows, mode) when the kernel stage is to be sharded over several GPUs, else None: an explicit
    multi_gpu.similarity(...) call, or SIMILARIPY_AMD_DEVICES=0,1,... in the environment.

    mode "threads" (default): the library's own multi-device call (sp_knn_args.n_devices, ABI 5) — one process, one host thread
    per device, nothing spawned, every device-side stage of the single-GPU path kept.  mode "processes" (chunk_rows given, or
    SIMILARIPY_AMD_MULTI_GPU_MODE=processes): one worker process per GPU under torch.distributed, the slabs gathered over RCCL
    (multi_gpu.run_call) — what streams a 10M-user job in chunks."""
    if _MULTI_GPU_ROUTE:
        r = _MULTI_GPU_ROUTE[-1]
        return r.devices, r.chunk_rows, r.mode
    from .multi_gpu import devices_from_env
    d = devices_from_env()
    if d:
        c = os.environ.get("SIMILARIPY_AMD_CHUNK_ROWS", "")
        chunk = int(c) if c else None
        mode = os.environ.get("SIMILARIPY_AMD_MULTI_GPU_MODE", "") or ("processes" if chunk else "threads")
        return d, chunk, mode
    return None


def _s_plus_impl(matrix1, matrix2, weight_depop_matrix1, weight_depop_matrix2, p1, p2, a1, l1, l2, l3,
                 t1, t2, c1, c2, k, stabilized_shrink, bayesian_shrink, additive_shrink, threshold,
                 binary, target_rows, filter_cols, target_cols, verbose, format_output,
                 p3_alpha=None, p3_depop_beta=None):
    """prepare -> kernel -> assembly.  The stages the reference runs on the host around its kernel are left to the device
    where the call allows it: the transpose and the norms of the `matrix2=None` call, the search for stored zeros, the
    preprocessing of p3alpha / rp3beta (p3_alpha / p3_depop_beta: matrix1 is the RAW matrix then) and the CSR assembly."""
    _say(verbose if isinstance(verbose, bool) else False, "Preprocessing")
    args = (matrix1, matrix2, weight_depop_matrix1, weight_depop_matrix2, p1, p2, a1, l1, l2, l3,
            t1, t2, c1, c2, k, stabilized_shrink, bayesian_shrink, additive_shrink, threshold,
            binary, target_rows, filter_cols, target_cols, verbose, format_output)
    p3kw = dict(p3_alpha=p3_alpha, p3_depop_beta=p3_depop_beta)
    route = multi_gpu_route()
    devices = None
    if route is not None and route[2] == "threads":
        devices = list(range(route[0])) if isinstance(route[0], int) else [int(d) for d in route[0]]
        route = None
    if route is not None:
        # several GPUs: the host stages once, here; the kernel stage in one worker process per GPU (multi_gpu.run_call)
        if p3_alpha is not None:
            raise ValueError("the multi-GPU route takes preprocessed matrices (similarity.p3alpha / rp3beta do that)")
        from . import multi_gpu
        call = prepare(*args, m2_on_device=False, check_zeros=True)
        _say(verbose, f"Computing on devices {route[0]}")
        res = multi_gpu.run_call(call, route[0], format_output, chunk_rows=route[1])
        _say(verbose, "Done")
        return res
    # stored zeros: looked for on the device, where the data goes anyway (also under `binary` when the library writes the ones into
    # its own copies, SP_FLAG_BINARY; where prepare has to build the ones itself it checks on the host whatever is asked here)
    opts = dict(check_zeros=False, csc_direct=True, binary_on_device=True, m2_sorted_on_device=True, selectors_sorted_on_device=True)
    while True:
        call = prepare(*args, m2_on_device=True, norms_on_device=True, keep_on_device=True, **opts, **p3kw)
        # CSR results are assembled on the device whatever the order of target_rows (SP_FLAG_CSR_OUT: the stable counting sort of
        # coo_to_csr.h:28-71, repeats included)
        csr_out = format_output == 'csr' and call.n_targets > 0 and call.n_targets * call.k <= np.iinfo(np.int32).max
        if csr_out and call.n_targets > 4096 and not bool(np.all(call.targets[1:] > call.targets[:-1])) and int(np.bincount(call.targets).max()) > 1024:
            # (a row asked for thousands of times: the device assembly orders a row's slots with a one-thread insertion sort, fine for the
            # handful of repeats real calls have, quadratic here — the slots come back and the host assembles, coo_to_csr.h:28-71)
            csr_out = False
        if csr_out and devices is not None and len(devices) > 1 and call.n_targets > 1 and not bool(np.all(call.targets[1:] > call.targets[:-1])):
            csr_out = False          # (several devices assemble the CSR rows of their own slices: needs ascending targets; else the slots come back)
        _say(verbose, "Computing")
        try:
            # (the row id of every slot entry is known on the host: the library's helper threads write `rows` while the device works,
            # a third of the COO download is never made)
            out = run_hip(call, want_rows=(format_output != 'csr'), check_zeros=not opts["check_zeros"], csr_out=csr_out, devices=devices,
                          progress=bool(verbose) if isinstance(verbose, bool) else False)
            break
        except _abi.ExplicitZerosError:
            if opts["check_zeros"]:
                raise
            opts["check_zeros"] = True          # eliminate_zeros on the host (s_plus.pyx:210-211), then again
        except _abi.UnsortedRowsError as exc:
            if isinstance(exc, _abi.UnsortedSelectorError) and opts["selectors_sorted_on_device"]:      # (its own return code: SP_EUNSORTED_SELECTOR)
                opts["selectors_sorted_on_device"] = False   # a stale has_sorted_indices flag: the order is verified (and a copy sorted) here, then again
            elif call.m1_is_m2t and opts["csc_direct"]:
                opts["csc_direct"] = False          # matrix1.tocsr() on the host (s_plus.pyx:205-206), then again
            elif call.check_m2_sorted and opts["m2_sorted_on_device"]:
                opts["m2_sorted_on_device"] = False  # sort_indices() on a copy of matrix2 here (s_plus_utils.pyx:562), then again
            else:
                raise
    _say(verbose, f"Building {format_output} matrix")
    if csr_out:
        indptr, indices, data = out
        res = sp.csr_array((data, indices, indptr), shape=(call.n_rows_m1, call.n_output_cols), dtype=np.float32)
    else:
        rows, cols, values, counts = out
        res = finish(call, rows, cols, values, counts, format_output)
    _say(verbose, "Done")
    return res
