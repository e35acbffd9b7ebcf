"""GPU parity tests of the register-resident row kernel (csrc/sp_rowreg_kernel.hpp), the kernel BASELINE configs[1] runs on.

It replaces `compute_similarities_parallel` (s_plus.h:265-453) for rows of <= 64 m1 entries and 16..224 work items under a
monotone epilogue (dot product, cosine-type with the column term folded in), 1024-thread shape.  Every test compares the HIP
result, through the C ABI, with the CPU oracle port on the same inputs (tie-aware: `so.compare_topk`), and checks through the
phase counters that the rows were in fact served by this kernel (phase slot 8) — or, where the test says so, by the others.
"""
from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import splus_oracle as so
from similaripy_amd import _host
from similaripy_amd.workloads import fixed_degree_csr

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-7      # north_star: float32 values within 1e-5 relative
BIG = dict(threads_per_wg=1024, table_slots=16384)      # the shape the kernel belongs to (auto-chosen for heavy rows, forced here)


def _matrix(n_rows=60000, n_cols=4000, nnz_row=20, seed=31):
    """m2 = m.T: 4000 rows of ~300 entries (two items each): a target row has nnz_row segments, ~2 * nnz_row items, ~6000 products
    over 60000 output columns (few collisions: a sparse-kernel row)."""
    return fixed_degree_csr(n_rows, n_cols, nnz_row, seed)


def _run(call, **tuning):
    rows, cols, vals, counts, info = _host.run_hip(call, time_kernel=True, **tuning)
    return rows, cols, vals, counts, info


def _check(call, what, expect_rowreg=None, **tuning):
    rows, cols, vals, counts, info = _run(call, **tuning)
    k, n = call.k, call.n_targets
    got = so.canonical(rows, cols, vals, call.targets, k)
    want = so.canonical(*so.run_kernel(call, "port"), call.targets, k)
    so.compare_topk(got, want, k, rtol=RTOL, atol=ATOL, what=what)
    r, c, v = rows.reshape(n, k), cols.reshape(n, k), vals.reshape(n, k)
    pad = np.arange(k)[None, :] >= counts[:, None]
    assert not r[pad].any() and not c[pad].any() and not v[pad].any(), f"{what}: padding not zero"
    assert np.all(r[~pad] == np.broadcast_to(call.targets[:, None], r.shape)[~pad]), f"{what}: rows != target"
    pc = info["phase_cycles"]
    if expect_rowreg is not None:
        assert pc[8] == expect_rowreg, f"{what}: {pc[8]} rows on the register-resident kernel, expected {expect_rowreg} (sparse {pc[9]}, handed on {pc[10]})"
    return pc, info


@pytest.mark.parametrize("name,kw", [("dot", {}), ("cosine", dict(l2=1)), ("asym", dict(l2=1, c1=0.3, c2=0.7)),
                                     ("rp3like", dict(l3=1, weight_depop_matrix2="sum", p2=0.6))],
                         ids=["dot", "cosine", "asym", "rp3like"])
def test_rowreg_monotone_epilogues(name, kw):
    m = _matrix()
    t = np.arange(0, 60000, 17)
    call = _host.prepare(m, k=50, target_rows=t, **kw)
    pc, info = _check(call, "rowreg " + name, expect_rowreg=t.shape[0], **BIG)
    assert pc[10] == 0 and info["rowreg_kernel_ms"] > 0
    # the same rows through the bitmap kernel (A/B flag) and with the static schedule: same answer
    _check(call, "rowreg off " + name, expect_rowreg=0, no_rowreg=True, **BIG)
    _check(call, "rowreg static " + name, expect_rowreg=t.shape[0], static_sched=True, **BIG)


def test_rowreg_not_used_for_general_epilogues_or_large_k():
    m = _matrix()
    t = np.arange(0, 60000, 37)
    _check(_host.prepare(m, k=50, l1=1, t1=0.7, t2=0.3, target_rows=t), "tversky", expect_rowreg=0, **BIG)
    _check(_host.prepare(m, k=50, l1=0.5, l2=0.5, stabilized_shrink=5, target_rows=t), "s_plus hybrid", expect_rowreg=0, **BIG)
    _check(_host.prepare(m, k=225, l2=1, target_rows=t), "k beyond the first stage's rounds", expect_rowreg=0, **BIG)
    _check(_host.prepare(m, k=224, l2=1, target_rows=t), "largest k", expect_rowreg=t.shape[0], **BIG)
    _check(_host.prepare(m, k=1, l2=1, target_rows=t), "k=1", expect_rowreg=t.shape[0], **BIG)


def test_rowreg_tied_values_take_the_first_stage_fallback():
    """Binary data: every product of a segment has the same value, the first stage's cutoff is reached by all 4096 products
    (it does not fit), so only the first waves' items are accepted and the other waves carry their slot 0 into the next stage."""
    m = _matrix(seed=32)
    m.data[:] = 1.0
    t = np.arange(5, 60000, 23)
    for kw in ({}, dict(l2=1)):
        _check(_host.prepare(m, k=30, target_rows=t, **kw), f"binary {kw}", expect_rowreg=t.shape[0], **BIG)
    m.data[:] = np.random.default_rng(2).integers(1, 4, m.nnz).astype(np.float32)    # three levels
    _check(_host.prepare(m, k=30, l2=1, target_rows=t), "three levels", expect_rowreg=t.shape[0], **BIG)


def test_rowreg_signed_values_and_thresholds():
    m = _matrix(seed=33)
    m.data = (m.data - 0.4).astype(np.float32)
    t = np.arange(3, 60000, 29)
    _check(_host.prepare(m, k=25, l2=1, target_rows=t), "signed cosine", expect_rowreg=t.shape[0], **BIG)
    _check(_host.prepare(m, k=25, threshold=0.05, target_rows=t), "signed dot, positive threshold", expect_rowreg=t.shape[0], **BIG)
    _check(_host.prepare(m, k=25, l2=1, threshold=-0.05, target_rows=t), "signed cosine, negative threshold", expect_rowreg=t.shape[0], **BIG)
    _check(_host.prepare(m, k=25, l2=1, threshold=0.2, target_rows=t), "threshold above most values", expect_rowreg=t.shape[0], **BIG)


def test_rowreg_ragged_rows_are_split_between_the_kernels():
    """Rows of 1..90 entries over m2 rows of 0..1500 entries: fewer than 16 items, more than 224 items or more than 64 entries go
    to the other row kernels, the rest here; empty m2 rows leave lanes without a segment between lanes with one; item counts
    are not multiples of 16 and most items are partial."""
    rng = np.random.default_rng(34)
    n_rows, n_mid, n_cols = 30000, 3000, 300000
    deg = rng.integers(1, 91, n_rows)
    indptr = np.concatenate(([0], np.cumsum(deg))).astype(np.int32)
    indices = np.concatenate([np.sort(rng.choice(n_mid, d, replace=False)) for d in deg]).astype(np.int32)
    m1 = sp.csr_array((rng.random(indices.shape[0], dtype=np.float32) + 0.01, indices, indptr), shape=(n_rows, n_mid))
    deg2 = rng.integers(0, 1500, n_mid)
    deg2[rng.random(n_mid) < 0.05] = 0                       # empty m2 rows
    indptr2 = np.concatenate(([0], np.cumsum(deg2))).astype(np.int32)
    indices2 = np.concatenate([np.sort(rng.choice(n_cols, d, replace=False)) for d in deg2] + [np.zeros(0, np.int64)]).astype(np.int32)
    m2 = sp.csr_array((rng.random(indices2.shape[0], dtype=np.float32) + 0.01, indices2, indptr2), shape=(n_mid, n_cols))
    t = np.arange(0, n_rows, 7)
    for kw in ({}, dict(l2=1)):
        call = _host.prepare(m1, m2, k=40, target_rows=t, **kw)
        pc, _ = _check(call, f"ragged {kw}", **BIG)
        assert 0 < pc[8] < t.shape[0], pc[8:11]              # some rows here, some elsewhere


def test_rowreg_matrix_filter():
    """filter_cols as a per-row matrix (s_plus.h:159-171): the excluded columns are marked in the collision bitmap and dropped
    at the scan of the collision set (the user-scoring idiom of BASELINE configs[4] on a wide catalogue)."""
    m = _matrix(seed=35)
    t = np.arange(1, 60000, 31)
    rng = np.random.default_rng(5)
    # exclude, for every row, its own column, a handful of its actual neighbours and a few random columns
    mm = (m[t] @ m.T).tocsr()
    rows_f, cols_f = [], []
    for i, r in enumerate(t):
        nb = mm.indices[mm.indptr[i]:mm.indptr[i + 1]]
        pick = np.unique(np.concatenate(([r], rng.choice(nb, min(40, nb.shape[0]), replace=False), rng.integers(0, 60000, 10))))
        rows_f.append(np.full(pick.shape[0], r)); cols_f.append(pick)
    f = sp.csr_array((np.ones(sum(x.shape[0] for x in cols_f), np.float32), (np.concatenate(rows_f), np.concatenate(cols_f))), shape=(60000, 60000))
    for kw in ({}, dict(l2=1)):
        call = _host.prepare(m, k=50, target_rows=t, filter_cols=f, **kw)
        pc, _ = _check(call, f"matrix filter {kw}", expect_rowreg=t.shape[0], **BIG)
        rows, cols, vals, counts = _host.run_hip(call, **BIG)
        kept = sp.csr_array((np.ones(int(counts.sum()), np.float32), (np.repeat(call.targets, counts), cols.reshape(-1, 50)[np.arange(50)[None, :] < counts[:, None]])), shape=(60000, 60000))
        assert kept.multiply(f).nnz == 0, "an excluded column was returned"


def test_rowreg_headline_shape_slice():
    """BASELINE configs[1] itself (1M x 100k, 64 per row, k = 100), a slice of the target rows: the kernel is chosen without
    any tuning argument and serves every row."""
    m = fixed_degree_csr(1_000_000, 100_000, 64, 12345)
    t = np.arange(11, 1_000_000, 1999)
    call = _host.prepare(m, k=100, l2=1, c1=0.5, c2=0.5, target_rows=t)
    pc, info = _check(call, "C2 slice", expect_rowreg=t.shape[0])
    assert pc[10] == 0
