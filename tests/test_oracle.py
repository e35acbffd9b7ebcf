"""CPU tier: pin the oracle.

1. oracle/libsplus_port.so (our C restatement of s_plus.h) == oracle/_ref/libsplus_ref.so (the
   reference header compiled in place), bit for bit, blocked and unblocked.
2. host logic + oracle port reproduce every golden vector produced by the imported reference.
3. oracle port agrees with the float64 dense definition.
"""
from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

import cases as C
from oracle import splus_oracle as so
from similaripy_amd import _host
import similaripy_amd as sim


def _rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32,
                           random_state=np.random.default_rng(seed))


KERNEL_PARAMS = [
    ("dot", {}),
    ("cosine", dict(l2=1)),
    ("asym", dict(l2=1, c1=0.2, c2=0.8)),
    ("tversky", dict(l1=1, t1=0.8, t2=0.4)),
    ("splus", dict(l1=0.5, l2=0.5, l3=1, weight_depop_matrix2="sum", stabilized_shrink=10)),
    ("pow_bayes", dict(l2=1, a1=0.7, bayesian_shrink=3)),
    ("thr", dict(l2=1, threshold=0.08)),
]


@pytest.mark.skipif(not so.available("reference"), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name,kw", KERNEL_PARAMS, ids=[p[0] for p in KERNEL_PARAMS])
@pytest.mark.parametrize("block_size", [0, 64, 300])
def test_port_equals_reference_kernel(name, kw, block_size):
    m = _rand((700, 500), 0.03, 5)
    call = _host.prepare(m, k=40, **kw)
    a = so.canonical(*so.run_kernel(call, "port", block_size=block_size), call.targets, call.k)
    b = so.canonical(*so.run_kernel(call, "reference", block_size=block_size), call.targets, call.k)
    for (ac, av), (bc, bv) in zip(a, b):
        np.testing.assert_array_equal(ac, bc)
        np.testing.assert_array_equal(av, bv)   # bit-exact: same order of float operations


@pytest.mark.skipif(not so.available("reference"), reason="oracle/_ref not built (needs /root/reference)")
def test_port_equals_reference_kernel_matrix_selectors():
    urm = _rand((120, 260), 0.05, 1)
    w = _rand((260, 260), 0.5, 2)
    for kw in (dict(filter_cols=urm), dict(target_cols=urm), dict(filter_cols=urm, target_rows=[5, 3, 100])):
        call = _host.prepare(urm, w, k=30, **kw)
        for bs in (0, 100):
            a = so.canonical(*so.run_kernel(call, "port", block_size=bs), call.targets, call.k)
            b = so.canonical(*so.run_kernel(call, "reference", block_size=bs), call.targets, call.k)
            for (ac, av), (bc, bv) in zip(a, b):
                np.testing.assert_array_equal(ac, bc)
                np.testing.assert_array_equal(av, bv)


def _our_canonical(res, targets, k):
    rows, cols, vals = res.row.astype(np.int32), res.col.astype(np.int32), res.data.astype(np.float32)
    return so.canonical(rows, cols, vals, targets, k), so.slot_counts(rows, cols, vals, targets, k)[0]


ALL_CASES = C.build_cases()


@pytest.mark.parametrize("case", ALL_CASES, ids=[c["name"] for c in ALL_CASES])
def test_host_logic_plus_oracle_match_golden(case, golden, oracle_backend):
    """wrapper -> prepare -> (oracle kernel) -> finish  ==  the reference's output."""
    m1, kw = C.call_kwargs(case, golden.inputs)
    entry = golden.entries[case["name"]]
    res = getattr(sim, case["fn"])(m1, **kw)
    assert res.dtype == np.float32 and list(res.shape) == entry["shape"]
    k = entry["k_eff"]
    if entry["format"] == "coo":
        assert isinstance(res, sp.coo_array)
        assert res.nnz == entry["stored_nnz"]          # padding is part of the COO
        targets = np.asarray(kw.get("target_rows", np.arange(m1.shape[0])), dtype=np.int32)
        got, got_counts = _our_canonical(res, targets, k)
        want, want_counts = golden.expected(case["name"])
        np.testing.assert_array_equal(got_counts, want_counts)
        so.compare_topk(got, want, k, rtol=1e-6, atol=1e-9, what=case["name"])
    else:
        assert isinstance(res, sp.csr_array)
        assert res.nnz == entry["stored_nnz"]
        r = res.copy()
        r.sort_indices()
        np.testing.assert_array_equal(r.indptr, golden.z[f"out/{case['name']}/indptr"])
        np.testing.assert_array_equal(r.indices, golden.z[f"out/{case['name']}/cols"])
        np.testing.assert_allclose(r.data, golden.z[f"out/{case['name']}/vals"], rtol=1e-6)


@pytest.mark.parametrize("name,kw,dense_kw", [
    ("dot", {}, {}),
    ("cosine", dict(l2=1), dict(l2=1)),
    ("tversky", dict(l1=1, t1=0.8, t2=0.4), dict(l1=1, t1=0.8, t2=0.4)),
    ("cos_add", dict(l2=1, additive_shrink=4.0), dict(l2=1, additive=4.0)),
    ("cos_bayes", dict(l2=1, bayesian_shrink=4.0), dict(l2=1, bayesian=4.0)),
    ("binary_jaccard", dict(l1=1, binary=True), dict(l1=1, binary=True)),
])
def test_port_matches_dense_definition(name, kw, dense_kw):
    m = _rand((400, 300), 0.04, 9)
    k = 25
    call = _host.prepare(m, k=k, **kw)
    got = so.canonical(*so.run_kernel(call, "port"), call.targets, k)
    S, mask = so.dense_similarity(m, **dense_kw)
    so.compare_topk(got, so.dense_topk(S, mask, k), k, rtol=3e-5, atol=1e-7, what=name)


# ------------------------------------------------------------------------------------------------------------
# two documented behaviours, pinned by reference-generated fixtures (tests/golden/make_quirks_golden.py)
# ------------------------------------------------------------------------------------------------------------
def _quirk_csr(z, name):
    return sp.csr_array((z[f"in/{name}/data"], z[f"in/{name}/indices"], z[f"in/{name}/indptr"]), shape=tuple(int(x) for x in z[f"in/{name}/shape"]))


def _sorted_triples(row, col, val):
    o = np.lexsort((val, col, row))
    return row[o], col[o], val[o]


@pytest.mark.parametrize("name,fn,kw", [("dup_dot", "dot_product", {}), ("dup_cosine", "cosine", {}), ("dup_dot_thr", "dot_product", dict(threshold=0.5)),
                                        ("dup_jaccard_shrink", "jaccard", dict(shrink=1.0))])
def test_port_reproduces_the_duplicate_listing_quirk(name, fn, kw, oracle_backend):
    """s_plus.h:112-117: a column whose running sum is exactly 0 when its next product arrives is listed a second time.  The ORACLE
    (a restatement of the reference) has the quirk too — stored triples equal the reference's, duplicates included; the HIP kernels
    do not reproduce it (tests/test_hip_parity.py::test_duplicate_listing_quirk_of_the_reference_is_not_reproduced)."""
    z = np.load(C.__file__.replace("cases.py", "quirks_golden.npz"))
    res = getattr(sim, fn)(_quirk_csr(z, "dup_m1"), _quirk_csr(z, "dup_m2"), k=8, verbose=False, **kw)
    got = _sorted_triples(res.row.astype(np.int32), res.col.astype(np.int32), res.data.astype(np.float32))
    want = _sorted_triples(z[f"out/{name}/row"], z[f"out/{name}/col"], z[f"out/{name}/val"])
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_allclose(got[2], want[2], rtol=1e-6)
    if "thr" not in name:
        r1 = want[1][want[0] == 1]
        assert np.count_nonzero(r1 == 2) == 2, "the fixture's row 1 lists column 2 twice"


@pytest.mark.parametrize("name,fn,kw", [("p3_alpha4", "p3alpha", dict(alpha=4.0)), ("rp3_alpha4_beta", "rp3beta", dict(alpha=4.0, beta=0.3))])
def test_p3_underflow_takes_the_host_statement(name, fn, kw, oracle_backend):
    """`data ** alpha` underflows to 0.0 for some entries: the reference drops them before its kernel runs.  The device-side preprocessing
    reports them (SP_EUNDERFLOW -> _abi.P3UnderflowError, emulated by the fixture) and the wrapper falls back on the host statement."""
    z = np.load(C.__file__.replace("cases.py", "quirks_golden.npz"))
    res = getattr(sim, fn)(_quirk_csr(z, "p3_m"), k=12, verbose=False, **kw)
    got = _sorted_triples(res.row.astype(np.int32), res.col.astype(np.int32), res.data.astype(np.float32))
    want = _sorted_triples(z[f"out/{name}/row"], z[f"out/{name}/col"], z[f"out/{name}/val"])
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_allclose(got[2], want[2], rtol=2e-6, atol=1e-37)
