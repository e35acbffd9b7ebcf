"""Shared fixtures.

Markers
-------
gpu   the test needs a real MI355X: it goes through libsimilaripy_hip.so (HIP kernels).
      Everything else runs on CPU: oracle vs golden vectors, host logic, ABI loading, gloo.

The oracle (oracle/) is the checker in both tiers; it is never what is being shipped.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
GOLDEN_DIR = ROOT / "tests" / "golden"
for p in (str(ROOT), str(GOLDEN_DIR)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (gfx950); runs the HIP kernels through the C ABI")


def _has_gpu() -> bool:
    try:
        from similaripy_amd import _abi
        return _abi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def has_gpu() -> bool:
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently: leave the tests alone.
    # A plain `pytest tests/` run on the CPU container skips them.
    if config.getoption("-m"):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device (run with -m gpu on an MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self):
        self.z = np.load(GOLDEN_DIR / "splus_golden.npz")
        self.manifest = json.loads((GOLDEN_DIR / "splus_golden_manifest.json").read_text())
        self.entries = {e["name"]: e for e in self.manifest["cases"]}
        self.inputs = {}
        names = sorted({k.split("/")[1] for k in self.z.files if k.startswith("in/")})
        for n in names:
            if f"in/{n}/array" in self.z.files:
                self.inputs[n] = self.z[f"in/{n}/array"]
            else:
                shape = tuple(int(x) for x in self.z[f"in/{n}/shape"])
                self.inputs[n] = sp.csr_array(
                    (self.z[f"in/{n}/data"], self.z[f"in/{n}/indices"], self.z[f"in/{n}/indptr"]), shape=shape)

    def expected(self, name):
        """canonical per-slot list [(cols, vals)] of a COO case, plus counts."""
        counts = self.z[f"out/{name}/counts"]
        cols, vals = self.z[f"out/{name}/cols"], self.z[f"out/{name}/vals"]
        off = np.concatenate(([0], np.cumsum(counts)))
        return [(cols[off[i]:off[i + 1]], vals[off[i]:off[i + 1]]) for i in range(counts.shape[0])], counts


@pytest.fixture(scope="session")
def golden() -> Golden:
    return Golden()


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Route similaripy_amd's kernel call to the CPU oracle port — CPU-tier tests only, to pin the
    HOST logic (wrappers, preprocessing, output assembly) against the reference's golden vectors.
    The product never does this."""
    from oracle import splus_oracle as so
    from similaripy_amd import _abi, _host

    def run(call, *a, **kw):
        import dataclasses
        import scipy.sparse as sp
        from oracle import norm_oracle
        explicit_m2 = not call.m1_is_m2t and not call.m2_is_m1t
        if call.m1_is_m2t:
            # SP_FLAG_M1_IS_M2_T: matrix1 came as CSC; the reference converts it with scipy (s_plus.pyx:205-206)
            if call.m2_indices.shape[0] > 1 and not _host._rows_sorted(call.m2_indices, call.m2_indptr):
                raise _abi.UnsortedRowsError("unsorted rows")
            m1 = sp.csr_array((call.m2_data, call.m2_indices, call.m2_indptr), shape=(call.n_rows_m2, call.n_rows_m1)).T.tocsr()
            call = dataclasses.replace(call, m1_data=np.ascontiguousarray(m1.data, dtype=np.float32), m1_indices=np.ascontiguousarray(m1.indices, dtype=np.int32),
                                       m1_indptr=np.ascontiguousarray(m1.indptr, dtype=np.int32), m1_is_m2t=False, m2_is_m1t=call.p3_alpha is not None)
        # (the order of run_host: zero count on the caller's values, order inside the rows of an explicit m2, ones, norms)
        if kw.get("check_zeros") and (np.count_nonzero(call.m1_data) != call.m1_data.shape[0] or np.count_nonzero(call.m2_data) != call.m2_data.shape[0]):
            raise _abi.ExplicitZerosError("stored zeros")            # what SP_FLAG_CHECK_ZEROS reports
        for mode, ip, ix in ((call.filter_mode, call.filter_m_indptr, call.filter_m_indices), (call.target_col_mode, call.target_col_m_indptr, call.target_col_m_indices)):
            # run_host looks at the order inside the rows of every MATRIX selector (sp_rows_sorted_kernel)
            if mode == _host.MODE_MATRIX and ix.shape[0] > 1 and not _host._rows_sorted(ix, ip):
                raise _abi.UnsortedSelectorError("MATRIX selector: rows do not have ascending column ids")
        if call.check_m2_sorted:
            # SP_FLAG_CHECK_SORTED: a descent inside a row of the explicit m2 goes back to the caller
            if call.m2_indices.shape[0] > 1 and not _host._rows_sorted(call.m2_indices, call.m2_indptr):
                raise _abi.UnsortedRowsError("unsorted rows")
            call = dataclasses.replace(call, check_m2_sorted=False)
        if call.binary_on_device:
            # SP_FLAG_BINARY: ones in the library's copies (s_plus.pyx:214-217)
            call = dataclasses.replace(call, m1_data=np.ones_like(call.m1_data), m2_data=np.ones_like(call.m2_data), binary_on_device=False)
        if call.norms_on_device is not None:
            # SP_FLAG_NORMS_ON_DEVICE: s_plus_utils.pyx:169-228 from the rows of m1 (explicit m2: rows of m1, columns of m2)
            c1, c2, add = call.norms_on_device
            if explicit_m2:
                sq1, sq2 = _host.build_squared_norms(call.m1_data, call.m1_indices, call.m1_indptr, call.n_rows_m2,
                                                     call.m2_data, call.m2_indices, call.m2_indptr, call.n_output_cols)
            else:
                sq1, sq2 = _host.build_squared_norms_m1t(call.m1_data, call.m1_indptr)
            rep = {}
            if call.l1 != 0:
                rep.update(Xtversky=sq1, Ytversky=sq2)
            if call.l2 != 0:
                xc, yc = _host.build_cosine_normalization(sq1, sq2, c1, c2, add)
                rep.update(Xcosine=xc, Ycosine=yc)
            call = dataclasses.replace(call, norms_on_device=None, **rep)
        if call.p3_alpha is not None:
            # SP_FLAG_P3_PREP / SP_FLAG_DEPOP_ROWSUM, as the reference does it on the host (similarity.py:410-415, 477-483)
            m1 = sp.csr_array((call.m1_data, call.m1_indices, call.m1_indptr), shape=(call.n_rows_m1, call.n_rows_m2))
            m2 = m1.T.tocsr()
            m2.sort_indices()
            rep = {}
            if call.depop_rowsum_p2 is not None:
                pop = np.asarray(m2.sum(axis=0)).ravel()
                rep["Ydepop"] = np.power(pop, np.float32(call.depop_rowsum_p2), dtype=np.float32)
            a1 = norm_oracle.normalize(m1, norm="l1")
            a1.data = np.power(a1.data, np.float32(call.p3_alpha))
            b1 = norm_oracle.normalize(m2, norm="l1")
            b1.data = np.power(b1.data, np.float32(call.p3_alpha))
            if np.count_nonzero(a1.data) != a1.data.shape[0] or np.count_nonzero(b1.data) != b1.data.shape[0]:
                raise _abi.P3UnderflowError("entries underflowed to 0.0")      # what the library reports (SP_EUNDERFLOW)
            call = dataclasses.replace(call, m1_data=np.ascontiguousarray(a1.data, dtype=np.float32), m2_data=np.ascontiguousarray(b1.data, dtype=np.float32),
                                       m2_indices=np.ascontiguousarray(b1.indices, dtype=np.int32), m2_indptr=np.ascontiguousarray(b1.indptr, dtype=np.int32),
                                       m2_is_m1t=False, p3_alpha=None, depop_rowsum_p2=None, **rep)
        if call.m2_is_m1t:
            # the product leaves m1^T to the device (SP_FLAG_M2_IS_M1_T); the oracle gets it from scipy, as the reference does
            m2 = sp.csr_array((call.m1_data, call.m1_indices, call.m1_indptr), shape=(call.n_rows_m1, call.n_rows_m2)).T.tocsr()
            m2.sort_indices()
            d2, i2, p2 = (np.ascontiguousarray(m2.data, dtype=np.float32), np.ascontiguousarray(m2.indices, dtype=np.int32),
                          np.ascontiguousarray(m2.indptr, dtype=np.int32))
            if call.col_keep is not None:
                # sp_knn_args.col_keep: the reference's _filter_matrix_columns on that m2 (s_plus_utils.pyx:424-490)
                d2, i2, p2 = _host.filter_matrix_columns(d2, i2, p2, call.n_output_cols, np.flatnonzero(call.col_keep).astype(np.int32))
            call = dataclasses.replace(call, m2_data=d2, m2_indices=i2, m2_indptr=p2, m2_is_m1t=False, col_keep=None)
        if call.col_keep is not None:
            # ... and on an explicit matrix2 (host mode compacts the uploaded copy)
            d2, i2, p2 = _host.filter_matrix_columns(call.m2_data, call.m2_indices, call.m2_indptr, call.n_output_cols, np.flatnonzero(call.col_keep).astype(np.int32))
            call = dataclasses.replace(call, m2_data=d2, m2_indices=i2, m2_indptr=p2, col_keep=None)
        rows, cols, values = so.run_kernel(call, "port")
        counts, _ = so.slot_counts(rows, cols, values, call.targets, call.k) if call.n_targets else (np.zeros(0, np.int32), None)
        if kw.get("csr_out"):
            # SP_FLAG_CSR_OUT: what the device assembles is what build_csr assembles on the host
            res = _host.build_csr(call.targets, cols, values, counts, call.k, call.n_rows_m1, call.n_output_cols)
            return res.indptr.astype(np.int32), res.indices.astype(np.int32), res.data.astype(np.float32)
        return (rows if kw.get("want_rows", True) else None), cols, values, counts

    monkeypatch.setattr(_host, "run_hip", run)
    monkeypatch.setattr(_host, "squared_norms_hip", lambda d1, p1, d2, i2, nc: (
        _host.csr_sum(np.square(d1, dtype=np.float32), None, p1, 0, axis=1), _host.csr_sum(np.square(d2, dtype=np.float32), i2, None, nc, axis=0)))
    monkeypatch.setattr(_host, "col_sums_hip", lambda d, i, nc, square, device=None: _host.csr_sum(np.square(d, dtype=np.float32) if square else d, i, None, nc, axis=0))
    # (the product's row normalisers run on the device: their NumPy restatement stands in)
    from oracle import norm_oracle
    from similaripy_amd import normalization
    monkeypatch.setattr(normalization, "_run", norm_oracle.inplace_run)
    # (same for the device-side norms of the `matrix2=None` call: their NumPy statement stands in)
    monkeypatch.setattr(_host, "squared_norms_m1t_hip", lambda data, indptr, device=None: _host.build_squared_norms_m1t(data, indptr))
    return run
