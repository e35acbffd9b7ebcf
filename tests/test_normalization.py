"""Row normalisers: normalize l1/l2/max, tf-idf, BM25, BM25+.

CPU tier  the NumPy restatement (oracle/norm_oracle.py) against golden vectors produced by the imported reference
          (tests/golden/make_norm_golden.py): pins the oracle.
GPU tier  the product (similaripy_amd.normalization -> sp_csr_normalize -> HIP segmented kernels) against the same
          golden vectors and against the oracle, plus the reference's documented flow bm25 -> cosine -> dot_product
          (tests/test_similarity.py:359-380, README.md:80-94).

Tolerance: the reference's row sums are sequential float loops that its compiler reorders (-ffast-math); here they are
NumPy reduceat (oracle) / wave reductions (device): values agree to a few ulp, RTOL below; the structure is exact.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "golden"))

import norm_cases as NC                                # noqa: E402
from oracle import norm_oracle                          # noqa: E402

RTOL32, RTOL64 = 3e-6, 1e-12
CASES = NC.build_cases()


@pytest.fixture(scope="module")
def norm_golden():
    z = np.load(ROOT / "tests" / "golden" / "norm_golden.npz")
    inputs = {n: sp.csr_array((z[f"in/{n}/data"], z[f"in/{n}/indices"], z[f"in/{n}/indptr"]), shape=tuple(z[f"in/{n}/shape"])) for n in ("A", "B", "C", "D")}
    # the seeded generator reproduces the stored inputs (numpy / scipy versions of the fixture: see splus_golden_manifest.json)
    return z, inputs


def _check(res, z, name, in_dtype):
    assert isinstance(res, sp.csr_array)
    res = res.copy()
    res.sort_indices()
    want_dtype = in_dtype if in_dtype in (np.float32, np.float64) else np.float32      # normalization.py:38-39
    assert res.data.dtype == want_dtype
    np.testing.assert_array_equal(res.indptr, z[f"out/{name}/indptr"])
    np.testing.assert_array_equal(res.indices, z[f"out/{name}/indices"])
    want = z[f"out/{name}/data"]
    assert want.dtype == want_dtype
    np.testing.assert_allclose(res.data, want, rtol=RTOL64 if want_dtype == np.float64 else RTOL32, atol=0, err_msg=name)


@pytest.mark.parametrize("name,fn,inp,kw", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_golden(norm_golden, name, fn, inp, kw):
    z, inputs = norm_golden
    m = inputs[inp]
    before = m.copy()
    res = getattr(norm_oracle, fn)(m, **kw)
    assert (before != m).nnz == 0                        # inplace=False
    _check(res, z, name, m.data.dtype)


def test_argument_errors_match_reference():
    """normalization.py:23-86: same exception types for the same mistakes, before any device work."""
    import similaripy_amd as sim
    m = sp.random_array((10, 8), density=0.3, format="csr", dtype=np.float32, random_state=np.random.default_rng(0))
    for mod in (sim, norm_oracle):
        with pytest.raises(ValueError):
            mod.normalize(m, norm="l3")
        with pytest.raises(ValueError):
            mod.normalize(m, axis=2)
        with pytest.raises(TypeError):
            mod.normalize(np.zeros((3, 3)))
        with pytest.raises(ValueError, match="tf_mode"):
            mod.tfidf(m, tf_mode="nope")
        with pytest.raises(ValueError, match="idf_mode"):
            mod.bm25(m, idf_mode="nope")
        with pytest.raises(ValueError, match="idf_mode"):
            mod.bm25plus(m, idf_mode="nope")


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,fn,inp,kw", CASES, ids=[c[0] for c in CASES])
def test_device_matches_reference_golden(norm_golden, name, fn, inp, kw):
    import similaripy_amd as sim
    z, inputs = norm_golden
    m = inputs[inp]
    before = m.copy()
    res = getattr(sim, fn)(m, **kw)
    assert (before != m).nnz == 0 and not np.shares_memory(res.data, m.data)
    _check(res, z, name, m.data.dtype)


@pytest.mark.gpu
def test_device_inplace_long_rows_and_empty_matrix():
    import similaripy_amd as sim
    rng = np.random.default_rng(5)
    # rows far longer than a wave (one item rated by 10^5 users), empty rows between them
    lens = np.array([0, 100_000, 3, 0, 65, 64, 63, 1, 20_000, 0], dtype=np.int64)
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    n_cols = 120_000
    indices = np.concatenate([np.sort(rng.choice(n_cols, n, replace=False)) for n in lens]).astype(np.int32)
    data = (rng.random(indptr[-1], dtype=np.float32) + 0.1)
    m = sp.csr_array((data, indices, indptr), shape=(len(lens), n_cols))
    for fn, kw in (("normalize", dict(norm="l1")), ("normalize", dict(norm="l2")), ("normalize", dict(norm="max")),
                   ("bm25", {}), ("tfidf", {}), ("bm25plus", dict(delta=0.7))):
        got = getattr(sim, fn)(m, **kw)
        want = getattr(norm_oracle, fn)(m, **kw)
        np.testing.assert_array_equal(got.indices, want.indices)
        np.testing.assert_allclose(got.data, want.data, rtol=2e-5, err_msg=fn)      # 10^5-term float32 sums in another order
    # inplace=True works on the caller's arrays
    m2 = m.copy()
    out = sim.normalize(m2, norm="l1", inplace=True)
    assert np.shares_memory(out.data, m2.data)
    np.testing.assert_allclose(np.add.reduceat(np.abs(m2.data), indptr[:-1][lens > 0]), 1.0, rtol=1e-5)
    empty = sp.csr_array((7, 5), dtype=np.float32)
    assert sim.bm25(empty).nnz == 0 and sim.normalize(empty).shape == (7, 5)


@pytest.mark.gpu
def test_reference_example_flow():
    """tests/test_similarity.py:359-380 / README.md:80-94: bm25 -> cosine(urm.T, k=50) -> dot_product(urm, sim.T,
    target_rows=[1, 14, 8], filter_cols=urm) — every step on the GPU, against the oracle chain."""
    import similaripy_amd as sim
    from oracle import splus_oracle as so
    from similaripy_amd import _host
    urm = sp.random_array((1000, 2000), density=0.025, format="csr", dtype=np.float32, random_state=np.random.default_rng(42))
    urm_n = sim.bm25(urm)
    np.testing.assert_allclose(urm_n.data, norm_oracle.bm25(urm).data, rtol=RTOL32)
    model = sim.cosine(urm_n.T, k=50, verbose=False)
    assert isinstance(model, sp.coo_array) and model.shape == (2000, 2000) and model.nnz > 0
    rec = sim.dot_product(urm, model.T, k=20, target_rows=[1, 14, 8], filter_cols=urm, verbose=False)
    assert rec.shape == (1000, 2000)
    rec = rec.tocsr()
    for u in (1, 14, 8):
        assert rec[[u], :].nnz > 0
        assert not np.intersect1d(rec[[u], :].indices, urm[[u], :].indices).size      # nothing seen is recommended
    assert rec.nnz == sum(rec[[u], :].nnz for u in (1, 14, 8))
    # the scoring step against the oracle on the same model
    call = _host.prepare(urm, model.T.tocsr(), k=20, target_rows=[1, 14, 8], filter_cols=urm)
    want = so.canonical(*so.run_kernel(call, "port"), call.targets, 20)
    got = []
    for u in (1, 14, 8):
        r = rec[[u], :]
        o = np.argsort(r.indices)
        got.append((r.indices[o].astype(np.int32), r.data[o].astype(np.float32)))
    want = [(c[v != 0], v[v != 0]) for c, v in want]
    so.compare_topk(got, want, 20, rtol=1e-5, atol=1e-7, what="example flow")
