"""CPU tier: the Python mirror of the reference's host-side steps, against intermediates captured from
the reference itself (tests/golden, 'mid/*') and against its documented error behaviour."""
from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

import similaripy_amd as sim
from oracle import splus_oracle as so
from similaripy_amd import _host
from oracle.norm_oracle import normalize      # (the NumPy restatement; the product's runs on the device: tests/test_normalization.py)


def _csr(m):
    m = m.tocsr()
    return np.asarray(m.data, np.float32), np.asarray(m.indices, np.int32), np.asarray(m.indptr, np.int32)


def test_squared_norms_and_cosine_terms(golden):
    A = golden.inputs["A"]
    d1, i1, p1 = _csr(A)
    d2, i2, p2 = _csr(A.T)
    sq1, sq2 = _host.build_squared_norms(d1, i1, p1, A.shape[1], d2, i2, p2, A.shape[0])
    np.testing.assert_allclose(sq1, golden.z["mid/A/sq1"], rtol=1e-6)
    np.testing.assert_allclose(sq2, golden.z["mid/A/sq2"], rtol=1e-6)
    xc, yc = _host.build_cosine_normalization(sq1, sq2, 0.2, 0.8, 10.0)
    np.testing.assert_allclose(xc, golden.z["mid/A/xcos_c0.2_add10"], rtol=1e-6)
    np.testing.assert_allclose(yc, golden.z["mid/A/ycos_c0.8_add10"], rtol=1e-6)
    np.testing.assert_allclose(_host.csr_sum(d1, i1, p1, A.shape[1], 1), golden.z["mid/A/rowsum"], rtol=1e-6)
    np.testing.assert_allclose(_host.csr_sum(d1, i1, p1, A.shape[1], 0), golden.z["mid/A/colsum"], rtol=1e-6)


def test_depop_terms(golden):
    A = golden.inputs["A"]
    d1, i1, p1 = _csr(A)
    d2, i2, p2 = _csr(A.T)
    xd, yd = _host.build_depop_normalization((d1, i1, p1, A.shape[1]), (d2, i2, p2, A.shape[0]),
                                             A.shape[0], A.shape[0], 'sum', golden.inputs["pop2"], 0.7, 0.3)
    np.testing.assert_allclose(xd, golden.z["mid/A/xdep_sum_p0.7"], rtol=1e-6)
    np.testing.assert_allclose(yd, golden.z["mid/A/ydep_pop2_p0.3"], rtol=1e-6)
    ones, _ = _host.build_depop_normalization((d1, i1, p1, 0), (d2, i2, p2, 0), 7, 9, 'none', 'none', 3.0, 2.0)
    assert ones.dtype == np.float32 and ones.shape == (7,) and (ones == 1).all()
    with pytest.raises(ValueError):
        _host.build_depop_normalization((d1, i1, p1, 0), (d2, i2, p2, 0), 7, 9, 'bogus', 'none', 1, 1)


def test_array_selectors(golden):
    tc = _host.compute_target_columns([2, 3, 400, -1], [1, 2, 3, 4, 5, 6, 299, 300], 300)
    np.testing.assert_array_equal(tc, golden.z["mid/target_columns"])
    A = golden.inputs["A"]
    d2, i2, p2 = _csr(A.T)
    fd, fi, fp = _host.filter_matrix_columns(d2, i2, p2, 300, tc)
    np.testing.assert_array_equal(fp, golden.z["mid/AT_filtered/indptr"])
    np.testing.assert_array_equal(fi, golden.z["mid/AT_filtered/indices"])
    np.testing.assert_array_equal(fd, golden.z["mid/AT_filtered/data"])
    # both None / both matrices -> everything
    np.testing.assert_array_equal(_host.compute_target_columns(None, None, 5), np.arange(5))
    np.testing.assert_array_equal(_host.compute_target_columns(A, A, 5), np.arange(5))


def test_matrix_selector(golden):
    mode, ip, ix = _host.build_column_selector(golden.inputs["URM"])
    assert mode == golden.manifest["URM_selector_mode"] == _host.MODE_MATRIX
    np.testing.assert_array_equal(ip, golden.z["mid/URM_selector/indptr"])
    np.testing.assert_array_equal(ix, golden.z["mid/URM_selector/indices"])
    assert _host.build_column_selector(None)[0] == _host.MODE_NONE
    assert _host.build_column_selector([])[0] == _host.MODE_NONE
    assert _host.build_column_selector(sp.csr_array((3, 3)))[0] == _host.MODE_NONE      # sparse without data
    assert _host.build_column_selector([1, 2])[0] == _host.MODE_ARRAY
    assert _host.build_column_selector(np.array([4]))[0] == _host.MODE_ARRAY
    # explicit zeros go, rows come out sorted, and the caller's matrix is left as it was (s_plus_utils.pyx:326-333 works on a copy)
    dirty = sp.csr_array((np.array([1, 0, 2, 3], dtype=np.float32), np.array([3, 1, 0, 2], dtype=np.int32), np.array([0, 3, 4], dtype=np.int32)), shape=(2, 4))
    before = (dirty.data.copy(), dirty.indices.copy(), dirty.indptr.copy())
    mode, ip, ix = _host.build_column_selector(dirty)
    assert mode == _host.MODE_MATRIX and ip.tolist() == [0, 2, 3] and ix.tolist() == [0, 3, 2]
    for a, b in zip(before, (dirty.data, dirty.indices, dirty.indptr)):
        np.testing.assert_array_equal(a, b)
    # a canonical matrix is used where it lies (no copy of 64 M entries for an URM that filters itself)
    clean = sp.csr_array((np.ones(3, dtype=np.float32), np.array([0, 2, 1], dtype=np.int32), np.array([0, 2, 3], dtype=np.int32)), shape=(2, 4))
    assert np.shares_memory(_host.build_column_selector(clean)[2], clean.indices)


@pytest.mark.parametrize("norm", ["l1", "l2", "max"])
@pytest.mark.parametrize("axis", [0, 1])
def test_normalize_matches_reference(golden, norm, axis):
    r = normalize(golden.inputs["E"], norm=norm, axis=axis)
    r.sort_indices()
    np.testing.assert_array_equal(r.indptr, golden.z[f"mid/E_norm_{norm}_ax{axis}/indptr"])
    np.testing.assert_array_equal(r.indices, golden.z[f"mid/E_norm_{norm}_ax{axis}/indices"])
    np.testing.assert_allclose(r.data, golden.z[f"mid/E_norm_{norm}_ax{axis}/data"], rtol=2e-6)


def test_normalize_l1_reference_own_test():
    """tests/test_normalization.py:12-22 of the reference, restated."""
    X = sp.random_array((100, 50), density=0.05, format="csr", dtype=np.float32, random_state=np.random.default_rng(42))
    Xn = normalize(X, norm="l1")
    expected = X.copy()
    rs = np.asarray(expected.sum(axis=1)).ravel()
    rs[rs == 0] = 1
    expected.data /= np.repeat(rs, np.diff(expected.indptr))
    np.testing.assert_allclose(Xn.toarray(), expected.toarray(), rtol=1e-5)
    assert X is not Xn and not np.shares_memory(X.data, Xn.data)          # inplace=False copies
    with pytest.raises(ValueError):
        normalize(X, norm="l3")
    with pytest.raises(ValueError):
        normalize(X, axis=2)
    with pytest.raises(TypeError):
        normalize(np.zeros((3, 3)))


def test_validation_errors_match_reference():
    """s_plus_utils.pyx:19-125: same exception types for the same mistakes."""
    m = sp.random_array((20, 10), density=0.3, format="csr", dtype=np.float32, random_state=np.random.default_rng(1))
    with pytest.raises(TypeError):
        sim.cosine(m.toarray(), verbose=False)
    with pytest.raises(TypeError):
        sim.cosine(m, matrix2=m.T.toarray(), verbose=False)
    with pytest.raises(ValueError, match="Incompatible matrix shapes"):
        sim.cosine(m, matrix2=m, verbose=False)
    with pytest.raises(ValueError, match="k must be >= 1"):
        sim.cosine(m, k=0, verbose=False)
    with pytest.raises(ValueError):
        sim.s_plus(m, pop1=np.ones(3), verbose=False)
    with pytest.raises(ValueError):
        sim.s_plus(m, pop2="avg", l3=1, verbose=False)
    with pytest.raises(ValueError, match="target_rows length"):
        sim.cosine(m, target_rows=list(range(21)), verbose=False)
    with pytest.raises(TypeError):
        sim.cosine(m, filter_cols=(1, 2), verbose=False)
    with pytest.raises(ValueError, match="does not match expected"):
        sim.cosine(m, filter_cols=sp.csr_array(np.ones((3, 3), np.float32)), verbose=False)
    with pytest.raises(TypeError, match="verbose must be boolean"):
        sim.cosine(m, verbose=1)
    with pytest.raises(ValueError, match="format_output"):
        sim.cosine(m, format_output="csc", verbose=False)
    with pytest.raises(ValueError, match="shrink_type"):
        sim.cosine(m, shrink=1, shrink_type="nope", verbose=False)


def test_prepare_does_not_touch_callers_matrices():
    m = sp.random_array((40, 30), density=0.2, format="csr", dtype=np.float64, random_state=np.random.default_rng(2))
    m.data[5] = 0.0                                   # explicit zero
    before = (m.data.copy(), m.indices.copy(), m.indptr.copy())
    call = _host.prepare(m, k=7, l2=1, binary=True)
    assert m.data.dtype == np.float64 and np.array_equal(m.data, before[0]) and np.array_equal(m.indices, before[1])
    assert call.m1_data.dtype == np.float32 and (call.m1_data == 1).all()
    assert call.m1_data.shape[0] == m.nnz - 1          # the explicit zero is gone from the kernel's view
    assert call.k == 7 and call.n_output_cols == 40 and call.n_rows_m2 == 30


def test_k_clamped_and_targets_dtype():
    m = sp.random_array((12, 9), density=0.4, format="csr", dtype=np.float32, random_state=np.random.default_rng(3))
    call = _host.prepare(m, k=500, target_rows=[3, 1])
    assert call.k == 12 and call.targets.dtype == np.int32 and call.targets.tolist() == [3, 1]
    with pytest.raises(ValueError):
        _host.prepare(m, k=3, target_rows=[12])


def test_unsorted_m2_rows_get_sorted():
    m2 = sp.csr_array((np.array([1, 2, 3, 4], np.float32), np.array([5, 1, 3, 0], np.int32), np.array([0, 2, 4], np.int32)), shape=(2, 8))
    m1 = sp.csr_array(np.ones((3, 2), np.float32))
    call = _host.prepare(m1, m2, k=4)
    assert _host._rows_sorted(call.m2_indices, call.m2_indptr)
    np.testing.assert_array_equal(call.m2_indices, [1, 5, 0, 3])
    np.testing.assert_array_equal(call.m2_data, [2, 1, 4, 3])


def test_trailing_empty_rows_do_not_raise(oracle_backend):
    """The reference raises IndexError in csr_sum for a trailing empty row (np.add.reduceat index == nnz,
    s_plus_utils.pyx:154); here the row simply has no neighbours."""
    m = sp.random_array((30, 20), density=0.2, format="csr", dtype=np.float32, random_state=np.random.default_rng(4)).tolil()
    m[29, :] = 0
    m[0, :] = 0
    m = sp.csr_array(m.tocsr())
    res = sim.cosine(m, k=5, verbose=False, format_output="csr")
    assert res[[29], :].nnz == 0 and res[[0], :].nnz == 0
    S, mask = so.dense_similarity(m, l2=1)
    got = []
    for i in range(30):
        row = res[[i], :].tocoo()
        o = np.argsort(row.col)
        got.append((row.col[o].astype(np.int32), row.data[o].astype(np.float32)))
    so.compare_topk(got, so.dense_topk(S, mask, 5), 5, rtol=3e-5)


@pytest.mark.parametrize("fn,kw", [("cosine", {}), ("jaccard", {"binary": True}), ("tversky", {"alpha": 0.6, "beta": 0.3}), ("dot_product", {}),
                                   ("p3alpha", {"alpha": 0.7}), ("rp3beta", {"alpha": 0.7, "beta": 0.5}), ("s_plus", {"l2": 1.0, "shrink": 2.0, "shrink_type": "additive", "c1": 0.3, "c2": 0.7}),
                                   # depopularisation 'sum' weights with a CSC matrix1 (pop1='sum' crashed in prepare on the CSC route)
                                   ("s_plus", {"l1": 0.0, "l2": 0.0, "l3": 1.0, "pop1": "sum", "beta1": 0.5}),
                                   ("s_plus", {"l1": 0.0, "l2": 0.5, "l3": 0.5, "pop2": "sum", "beta2": 0.4}),
                                   ("s_plus", {"l1": 0.3, "l2": 0.3, "l3": 0.4, "pop1": "sum", "pop2": "sum", "beta1": 0.5, "beta2": 0.7})])
def test_csc_matrix1_takes_the_direct_route(fn, kw, oracle_backend):
    """A CSC matrix1 (`URM.T`) is handed over as the CSR of matrix2 (SP_FLAG_M1_IS_M2_T) with the norms left to the callee
    (SP_FLAG_NORMS_ON_DEVICE); the result is the one of the host-converted CSR call (s_plus.pyx:205-206)."""
    urm = sp.random_array((40, 55), density=0.15, format="csr", dtype=np.float32, random_state=np.random.default_rng(8))
    item = urm.T
    call = _host.prepare(item, k=6, l1=0.5, l2=0.5, m2_on_device=True, norms_on_device=True, csc_direct=True, c1=0.3, additive_shrink=1.0)
    assert call.m1_is_m2t and not call.m2_is_m1t and call.m1_indptr.size == 0 and call.Xcosine.size == 0
    assert call.norms_on_device == (float(np.float32(0.3)), 0.5, 1.0)
    np.testing.assert_array_equal(call.m2_indptr, urm.indptr)
    # without norms_on_device there is no way to take norms: the host conversion is used
    assert not _host.prepare(item, k=6, l2=1.0, m2_on_device=True, csc_direct=True).m1_is_m2t
    assert _host.prepare(item, k=6, m2_on_device=True, csc_direct=True).m1_is_m2t          # (dot product: no norms needed)
    a = getattr(sim, fn)(item, k=6, verbose=False, format_output="csr", **kw)
    b = getattr(sim, fn)(item.tocsr(), k=6, verbose=False, format_output="csr", **kw)
    assert a.shape == b.shape == (55, 55)
    np.testing.assert_allclose(a.toarray(), b.toarray(), rtol=1e-6, atol=0)
    # columns with descending row ids: reported by the callee, converted on the host
    perm = item.copy()
    for c in range(perm.shape[1]):
        lo, hi = perm.indptr[c], perm.indptr[c + 1]
        perm.indices[lo:hi] = perm.indices[lo:hi][::-1].copy()
        perm.data[lo:hi] = perm.data[lo:hi][::-1].copy()
    np.testing.assert_allclose(getattr(sim, fn)(perm, k=6, verbose=False, format_output="csr", **kw).toarray(), b.toarray(), rtol=1e-6, atol=0)


def test_array_selectors_and_depop_weights_keep_the_device_transpose(golden, monkeypatch):
    """ARRAY filter_cols / target_cols (compute_target_columns + _filter_matrix_columns, s_plus_utils.pyx:364-490) and the
    depopularisation weights (:231-278) no longer need m2 on the host: the call carries a column mask for the device-side
    transpose, and the 'sum' weights of m2 = m1^T come from the rows of m1 with np.bincount's arithmetic."""
    A = golden.inputs["A"]
    n = A.shape[0]
    fc, tc = list(range(0, 300, 3)), [1, 2, 3, 4, 5, 6, 50, 51, 250, 1000, -4]
    call = _host.prepare(A, k=10, filter_cols=fc, target_cols=tc, m2_on_device=True)
    assert call.m2_is_m1t and call.m2_data.size == 0 and call.col_keep.dtype == np.uint8 and call.col_keep.shape == (n,)
    np.testing.assert_array_equal(np.flatnonzero(call.col_keep), _host.compute_target_columns(fc, tc, n))
    assert _host.prepare(A, k=10, m2_on_device=True).col_keep is None
    # the host route (multi-GPU staging) still filters m2 itself
    host = _host.prepare(A, k=10, l2=1.0, filter_cols=fc, target_cols=tc, m2_on_device=False)
    assert not host.m2_is_m1t and host.col_keep is None and host.m2_data.size < A.nnz
    # weights: 'sum' of m2's columns == what the reference takes from the host transpose, bit for bit (the device's np.bincount,
    # sp_csr_col_sums_f32 over the row id of every entry of m1, is stood in for by its NumPy statement: no GPU in this tier)
    monkeypatch.setattr(_host, "col_sums_hip", lambda d, i, nc, square, device=None: _host.csr_sum(np.square(d, dtype=np.float32) if square else d, i, None, nc, axis=0))
    d1, i1, p1 = _csr(A)
    m2 = A.T.tocsr()
    d2, i2, p2 = _csr(m2)
    for w1, w2 in (("sum", "sum"), ("none", golden.inputs["pop2"]), (golden.inputs["pop1"], "none")):
        dev = _host.prepare(A, k=10, l3=1.0, weight_depop_matrix1=w1, weight_depop_matrix2=w2, p1=0.7, p2=0.3, m2_on_device=True)
        ref = _host.prepare(A, k=10, l3=1.0, weight_depop_matrix1=w1, weight_depop_matrix2=w2, p1=0.7, p2=0.3, m2_on_device=False)
        assert dev.m2_is_m1t and not ref.m2_is_m1t
        np.testing.assert_array_equal(dev.Xdepop, ref.Xdepop)
        np.testing.assert_array_equal(dev.Ydepop, ref.Ydepop)
    # p3alpha preprocessing inside the call carries the mask too (round 4): the library drops the columns from the NORMALISED m2
    p3 = _host.prepare(A, k=10, filter_cols=fc, m2_on_device=True, p3_alpha=1.0)
    assert p3.m2_is_m1t and p3.p3_alpha == 1.0 and p3.col_keep is not None and not p3.col_keep[fc].any()


@pytest.mark.parametrize("fn,kw", [("p3alpha", dict(alpha=0.8)), ("rp3beta", dict(alpha=0.8, beta=0.4)), ("rp3beta", dict(alpha=1.3, beta=0.6, shrink=2.0))])
def test_p3_with_array_selectors_equals_the_reference_order_of_operations(fn, kw, oracle_backend, golden):
    """p3alpha / rp3beta with ARRAY filter_cols / target_cols: the reference L1-normalises the rows of matrix2 FIRST and drops the
    columns afterwards (similarity.py:410-415, then s_plus_utils.pyx:424-490 inside s_plus).  The device route (SP_FLAG_P3_PREP with
    col_keep) must give what the host statement of that order gives: compared here through the oracle backend with the explicit
    `matrix2 = matrix1.T` call, which preprocesses on the host."""
    A = golden.inputs["A"]
    fc, tc = list(range(0, 300, 7)), list(range(5, 290))
    a = getattr(sim, fn)(A, k=8, filter_cols=fc, target_cols=tc, verbose=False, format_output="csr", **kw)
    b = getattr(sim, fn)(A, sp.csr_array(A.T), k=8, filter_cols=fc, target_cols=tc, verbose=False, format_output="csr", **kw)
    assert a.nnz == b.nnz and a.nnz > 0
    np.testing.assert_allclose(a.toarray(), b.toarray(), rtol=2e-6, atol=0)
    assert not a[:, fc].nnz and not a[:, [0, 1, 2, 3, 4, 295, 299]].nnz


def test_coo_attached_without_the_constructor_equals_the_constructed_one():
    rng = np.random.default_rng(0)
    rows = np.repeat(np.arange(50, dtype=np.int32), 4)
    cols = rng.integers(0, 70, 200).astype(np.int32)
    vals = rng.random(200, dtype=np.float32)
    a = _host.build_coo(rows, cols, vals, 50, 70)
    b = sp.coo_array((vals, (rows, cols)), shape=(50, 70), dtype=np.float32)
    assert isinstance(a, sp.coo_array) and a.shape == b.shape and a.nnz == b.nnz == 200 and a.dtype == np.float32
    assert a.row.dtype == b.row.dtype and np.array_equal(a.row, b.row) and np.array_equal(a.col, b.col) and np.array_equal(a.data, b.data)
    assert (a.tocsr() != b.tocsr()).nnz == 0 and np.allclose((a @ np.ones(70, np.float32)), (b @ np.ones(70, np.float32)))
    # arrays that are not what the kernel returns take the constructor
    c = _host.build_coo(rows.astype(np.int64), cols.astype(np.int64), vals, 50, 70)
    assert (c.tocsr() != b.tocsr()).nnz == 0


def test_csr_and_coo_assembly():
    targets = np.array([4, 1, 1], dtype=np.int32)           # unsorted, repeated
    k = 3
    cols = np.array([7, 2, 0, 5, 0, 0, 9, 8, 6], np.int32)
    vals = np.array([.5, .25, 0, .75, 0, 0, .1, 0.0, .3], np.float32)   # slot 2 holds a genuine zero value
    counts = np.array([2, 1, 3], np.int32)
    rows = np.array([4, 4, 0, 1, 0, 0, 1, 1, 1], np.int32)
    csr = _host.build_csr(targets, cols, vals, counts, k, 6, 10)
    assert isinstance(csr, sp.csr_array) and csr.shape == (6, 10) and csr.dtype == np.float32
    dense = np.zeros((6, 10), np.float32)
    dense[4, 7], dense[4, 2], dense[1, 5], dense[1, 9], dense[1, 6] = .5, .25, .75, .1, .3
    np.testing.assert_array_equal(csr.toarray(), dense)
    assert csr.nnz == 5                                       # padding and the genuine zero are eliminated
    coo = _host.build_coo(rows, cols, vals, 6, 10)
    assert isinstance(coo, sp.coo_array) and coo.nnz == 9     # padding kept
    np.testing.assert_array_equal(coo.toarray(), dense)


def test_csr_assembly_with_64_bit_indices():
    """build_csr_matrix picks 64-bit indices when n_targets * k or the column count exceed int32 (utils.pyx:141-173,
    coo_to_csr<long>): exercised with a column count beyond 2^31 on a tiny result."""
    targets = np.array([0, 2], dtype=np.int32)
    cols = np.array([5, 1, 0, 7, 0, 0], np.int32)
    vals = np.array([1.0, 2.0, 0.0, 3.0, 0.0, 0.0], np.float32)
    counts = np.array([2, 1], np.int32)
    n_cols = 2 ** 31 + 5
    csr = _host.build_csr(targets, cols, vals, counts, 3, 3, n_cols)
    assert csr.indices.dtype == np.int64 and csr.indptr.dtype == np.int64 and csr.shape == (3, n_cols)
    assert csr.nnz == 3 and csr.indptr.tolist() == [0, 2, 2, 3]
    assert sorted(zip(csr.indices[:2].tolist(), csr.data[:2].tolist())) == [(1, 2.0), (5, 1.0)] and csr.indices[2] == 7


def test_binary_values_stay_the_callers_when_the_device_writes_the_ones():
    """prepare(binary=True, binary_on_device=True): nothing on the host reads the values of a `matrix2=None` call with device-built norms, so
    they reach the boundary as they are (SP_FLAG_BINARY: ones written into the uploaded copies, s_plus.pyx:214-217); with depop weights, an
    explicit matrix2 or host-built norms the ones are made here — and then the stored zeros are removed here as well."""
    m = sp.random_array((300, 150), density=0.05, format="csr", dtype=np.float32, random_state=np.random.default_rng(1))
    c = _host.prepare(m, k=5, l2=1, binary=True, m2_on_device=True, norms_on_device=True, binary_on_device=True, check_zeros=False)
    assert c.binary_on_device and np.shares_memory(c.m1_data, m.data)
    c = _host.prepare(m, k=5, l2=1, l3=1, binary=True, m2_on_device=True, norms_on_device=True, binary_on_device=True, check_zeros=False)
    assert not c.binary_on_device and (c.m1_data == 1).all()
    c = _host.prepare(m, k=5, l2=1, binary=True)
    assert not c.binary_on_device and (c.m1_data == 1).all() and (c.m2_data == 1).all()
    z = m.copy()
    z.data[3] = 0
    c = _host.prepare(z, z.T.tocsr(), k=5, l2=1, binary=True, binary_on_device=True, check_zeros=False)
    assert not c.binary_on_device and c.m1_data.shape[0] == z.nnz - 1 and z.data[3] == 0


# ------------------------------------------------------------------------------------------------------------
# route table: what the public call asks of the library for every kind of input (VERDICT r4 #12 / ADVICE r3's crash site)
# ------------------------------------------------------------------------------------------------------------
def _flags_of_public_call(monkeypatch, fn, m1, **kw):
    """The SP_FLAG_* word (as names) the FIRST library call of a public wrapper carries, captured at the C boundary."""
    from similaripy_amd import _abi

    seen = {}

    class Stop(Exception):
        pass

    def fake_call(a):
        seen["flags"] = int(a.flags)
        seen["col_keep"] = bool(a.col_keep)
        raise Stop()

    monkeypatch.setattr(_abi, "call_knn", fake_call)
    monkeypatch.setattr(_abi, "require_device", lambda: 1)
    from oracle import norm_oracle
    from similaripy_amd import normalization
    monkeypatch.setattr(normalization, "_run", norm_oracle.inplace_run)      # (the host statement of p3alpha normalises through the device library)
    from similaripy_amd import _host
    monkeypatch.setattr(_host, "col_sums_hip", lambda d, i, nc, square, device=None: _host.csr_sum(np.square(d, dtype=np.float32) if square else d, i, None, nc, axis=0))
    with pytest.raises(Stop):
        getattr(sim, fn)(m1, verbose=False, **kw)
    names = {n[len("SP_FLAG_"):] for n in dir(_abi) if n.startswith("SP_FLAG_") and seen["flags"] & getattr(_abi, n)}
    return names - {"NO_ROWS_OUT"}, seen["col_keep"]


_RT_A = sp.random_array((40, 30), density=0.2, format="csr", dtype=np.float32, random_state=np.random.default_rng(1))
_RT_W = sp.random_array((30, 25), density=0.3, format="csr", dtype=np.float32, random_state=np.random.default_rng(2))
_RT_SEL = sp.random_array((40, 40), density=0.1, format="csr", dtype=np.float32, random_state=np.random.default_rng(3))
_ALWAYS = {"CHECK_ZEROS"}
ROUTES = [
    # (id, fn, matrix1, kwargs, expected flags beyond CHECK_ZEROS, col_keep set?)
    ("dot csr", "dot_product", _RT_A, {}, {"M2_IS_M1_T"}, False),
    ("cosine csr", "cosine", _RT_A, {}, {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, False),
    ("cosine csr out", "cosine", _RT_A, dict(format_output="csr"), {"M2_IS_M1_T", "NORMS_ON_DEVICE", "CSR_OUT"}, False),
    ("cosine csc", "cosine", _RT_A.tocsc(), {}, {"M1_IS_M2_T", "NORMS_ON_DEVICE"}, False),
    ("dot csc", "dot_product", _RT_A.tocsc(), {}, {"M1_IS_M2_T"}, False),
    ("jaccard binary", "jaccard", _RT_A, dict(binary=True), {"M2_IS_M1_T", "NORMS_ON_DEVICE", "BINARY"}, False),
    ("cosine csc binary", "cosine", _RT_A.tocsc(), dict(binary=True), {"M1_IS_M2_T", "NORMS_ON_DEVICE", "BINARY"}, False),
    ("cosine explicit m2", "cosine", _RT_A, dict(matrix2=_RT_W), {"NORMS_ON_DEVICE", "CHECK_SORTED"}, False),
    ("dot explicit m2 binary", "dot_product", _RT_A, dict(matrix2=_RT_W, binary=True), {"CHECK_SORTED", "BINARY"}, False),
    ("cosine filter list", "cosine", _RT_A, dict(filter_cols=[1, 2, 3]), {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, True),
    ("cosine csc + target list (no CSC route)", "cosine", _RT_A.tocsc(), dict(target_cols=[0, 5]), {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, True),
    ("explicit m2 + filter list", "cosine", _RT_A, dict(matrix2=_RT_W, filter_cols=[1, 2]), {"NORMS_ON_DEVICE", "CHECK_SORTED"}, True),
    ("cosine filter matrix", "cosine", _RT_A, dict(filter_cols=_RT_SEL), {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, False),
    ("p3alpha", "p3alpha", _RT_A, dict(alpha=0.8), {"M2_IS_M1_T", "P3_PREP"}, False),
    ("rp3beta", "rp3beta", _RT_A, dict(alpha=0.8, beta=0.4), {"M2_IS_M1_T", "P3_PREP", "DEPOP_ROWSUM"}, False),
    ("p3alpha csc", "p3alpha", _RT_A.tocsc(), dict(alpha=0.8), {"M1_IS_M2_T", "P3_PREP"}, False),
    ("p3alpha + filter list", "p3alpha", _RT_A, dict(alpha=0.8, filter_cols=[2]), {"M2_IS_M1_T", "P3_PREP"}, True),
    ("p3alpha explicit m2 (host statement)", "p3alpha", _RT_A, dict(matrix2=sp.csr_array(_RT_A.T)), {"CHECK_SORTED"}, False),
    ("s_plus pop2 sum", "s_plus", _RT_A, dict(l3=1.0, pop2="sum"), {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, False),
    ("s_plus pop1 sum csc (no CSC route)", "s_plus", _RT_A.tocsc(), dict(l3=1.0, pop1="sum"), {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, False),
    ("s_plus pop2 sum binary (ones made on the host)", "s_plus", _RT_A, dict(l3=1.0, pop2="sum", binary=True), {"M2_IS_M1_T", "NORMS_ON_DEVICE"}, False),
]


@pytest.mark.parametrize("rid,fn,m1,kw,want,keep", ROUTES, ids=[r[0] for r in ROUTES])
def test_route_table_public_call_to_library_flags(rid, fn, m1, kw, want, keep, monkeypatch):
    """Every legal combination of _host.prepare's route switches, written down as (input kind -> SP_FLAG_* set at the C boundary)."""
    got, got_keep = _flags_of_public_call(monkeypatch, fn, m1, **kw)
    # (CHECK_ZEROS rides on every first call — also where prepare made the ones of `binary` itself and has looked already: harmless)
    assert got == want | _ALWAYS, (rid, sorted(got))
    assert got_keep == keep, rid


def test_matrix_selector_with_a_stale_sorted_flag(oracle_backend):
    """scipy caches has_sorted_indices; an in-place edit of .indices leaves a stale True behind.  The library looks at the order itself
    (MATRIX selector rows: UnsortedRowsError), the wrapper then verifies and sorts a copy: the filter still filters."""
    urm = sp.random_array((60, 90), density=0.1, format="csr", dtype=np.float32, random_state=np.random.default_rng(5))
    w = sp.random_array((90, 90), density=0.4, format="csr", dtype=np.float32, random_state=np.random.default_rng(6))
    urm.sort_indices()
    stale = urm.copy()
    assert stale.has_sorted_indices      # (looked at once: scipy caches the answer)
    for r in range(stale.shape[0]):      # reverse every row in place; the cached flag still says "sorted"
        a, b = stale.indptr[r], stale.indptr[r + 1]
        stale.indices[a:b] = stale.indices[a:b][::-1].copy()
        stale.data[a:b] = stale.data[a:b][::-1].copy()
    assert stale.has_sorted_indices
    want = sim.dot_product(urm, w, k=20, filter_cols=urm, verbose=False, format_output="csr")
    got = sim.dot_product(urm, w, k=20, filter_cols=stale, verbose=False, format_output="csr")
    assert got.multiply(urm).nnz == 0
    assert (abs(got - want) > 1e-6).nnz == 0


def test_get_num_threads_counterpart(monkeypatch):
    """similaripy.cython_code.utils.get_num_threads (utils.pyx:18-25; tests/test_similarity.py:384-390) has a counterpart: the device count."""
    from similaripy_amd import _abi
    monkeypatch.setattr(_abi, "device_count", lambda: 3)
    assert sim.cython_code.utils.get_num_threads() == 3 and sim.get_num_threads() == 3 and sim.device_count() == 3
