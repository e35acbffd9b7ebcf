"""GPU tier, BASELINE.json configs[1] at FULL size (cosine, 1M x 100k, 64 nnz/row, k=100).

The oracle cannot finish 1M rows in seconds, so the full result is checked through size-independent
properties of cosine similarity, and a random sample of rows is compared with the oracle exactly
(tie-aware, 1e-5 relative)."""
from __future__ import annotations

import copy
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from bench import fixed_degree_csr                     # noqa: E402  (the canonical C2 generator)
from oracle import splus_oracle as so                  # noqa: E402
from similaripy_amd import _host                       # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    m = fixed_degree_csr(1_000_000, 100_000, 64, 12345)
    call = _host.prepare(m, k=100, l2=1, c1=0.5, c2=0.5)
    rows, cols, vals, counts = _host.run_hip(call)
    return m, call, rows.reshape(-1, 100), cols.reshape(-1, 100), vals.reshape(-1, 100), counts


def test_full_size_properties(c2):
    m, call, rows, cols, vals, counts = c2
    n, k = 1_000_000, 100
    # every row has far more than k candidates (~40k): all slots are used, no padding
    assert counts.min() == k and counts.max() == k
    assert np.array_equal(rows, np.broadcast_to(np.arange(n, dtype=np.int32)[:, None], (n, k)))
    assert cols.min() >= 0 and cols.max() < n
    # cosine of non-negative data: 0 < value <= 1 (+ float32 slack)
    assert vals.min() > 0.0 and vals.max() <= 1.0 + 2e-6
    # the row itself is always among its neighbours with similarity 1 (diagonal is kept)
    self_pos = (cols == np.arange(n, dtype=np.int32)[:, None])
    assert self_pos.sum(axis=1).min() == 1 and self_pos.sum(axis=1).max() == 1
    np.testing.assert_allclose(vals[self_pos], 1.0, rtol=1e-5)
    assert np.array_equal(vals.max(axis=1), vals[self_pos])
    # no column twice in a slot (spot check on 20k rows: a full check would sort 1e8 entries)
    pick = np.random.default_rng(0).choice(n, 20_000, replace=False)
    srt = np.sort(cols[pick], axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])
    # symmetry of cosine on m @ m.T: if j is in i's list with value v and i is in j's list, the values agree
    i_idx = pick[:2000]
    for i in i_idx[:200]:
        for j, v in zip(cols[i, :5], vals[i, :5]):
            hit = np.flatnonzero(cols[j] == i)
            if hit.size:
                assert abs(vals[j, hit[0]] - v) <= 1e-5 * max(abs(v), 1e-12)


def test_full_size_sample_vs_oracle(c2):
    m, call, rows, cols, vals, counts = c2
    k = 100
    sample = np.sort(np.random.default_rng(1).choice(1_000_000, 300, replace=False)).astype(np.int32)
    sub = copy.copy(call)
    sub.targets = sample
    want = so.canonical(*so.run_kernel(sub, "port"), sample, k)
    got = []
    for t in sample:
        o = np.argsort(cols[t], kind="stable")
        got.append((cols[t][o], vals[t][o]))
    so.compare_topk(got, want, k, rtol=1e-5, atol=1e-7, what="C2 sample")
