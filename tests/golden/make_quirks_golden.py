"""Generate tests/golden/quirks_golden.npz by running the REFERENCE implementation on two inputs built to hit
behaviours that DESIGN.md §2 documents ("What's weak" #3):

  dup_*   the duplicate-listing quirk of SparseMatrixMultiplier::add (s_plus.h:112-117): "running sum == 0" is taken for
          "first touch", so a column whose partial sum is exactly 0 when its next product arrives is listed twice and
          the reference emits it a second time with xy = 0.  Signed data, three products on one column: +2, -2, +3.
  p3_*    p3alpha / rp3beta on a matrix in which `data ** alpha` underflows to 0.0 for some stored entries
          (similarity.py:410-415); s_plus then removes them (eliminate_zeros, s_plus.pyx:210-211), i.e. they are no
          candidates in the reference.

Runs only in the build container:

    rm -rf /tmp/simref /tmp/simref_install && cp -r /root/reference /tmp/simref
    cd /tmp/simref && pip install --no-build-isolation --no-deps --no-index --target /tmp/simref_install .
    cd /tmp && PYTHONPATH=/tmp/simref_install:/root/repo python /root/repo/tests/golden/make_quirks_golden.py

The fixture holds DATA only (inputs and the reference's outputs as stored COO triples)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent

import similaripy as ref  # noqa: E402  (the reference, from /tmp/simref_install)

assert "/root/repo" not in ref.__file__, "this script must import the REFERENCE package"


def dup_inputs():
    # m1: 4 rows x 3 columns; m2: 3 rows x 8 columns.  Row 1 of m1 meets column 2 of m2 three times: +2, -2 (sum exactly 0), +3.
    # (row 1 / column 2: away from row 0 and column 0, whose (0, 0, 0.0) triples cannot be told from padding)
    m1 = sp.csr_array(np.array([[1.0, 0.0, 1.0],
                                [1.0, 1.0, 1.0],      # the quirk: column 2 receives +2, -2, +3
                                [0.5, 0.5, 0.0],      # +1 -1 on column 2, nothing after: sum 0, listed once, value 0
                                [0.0, 1.0, 1.0]], dtype=np.float32))
    m2 = sp.csr_array(np.array([[0.0, 1.0, 2.0, 0.0, 0.0, 0.0, 0.0, 0.0],
                                [0.0, 0.0, -2.0, 1.0, 0.0, 0.5, 0.0, 0.0],
                                [0.0, 0.0, 3.0, 0.0, 1.0, 0.0, 0.0, -1.0]], dtype=np.float32))
    return m1, m2


def p3_inputs():
    rng = np.random.default_rng(11)
    d = rng.random((12, 9)).astype(np.float32)
    d[d < 0.55] = 0.0
    d[0, :] = 0.0
    d[0, 0], d[0, 3], d[0, 5] = 1.0, 1e-12, 2e-13          # (x / sum) ** 4 underflows float32 for the two small ones
    d[4, 2] = 3e-14
    d[7, 7], d[7, 1] = 1e-13, 0.7
    return sp.csr_array(d)


def coo_triples(res):
    assert isinstance(res, sp.coo_array) and res.dtype == np.float32
    return res.row.astype(np.int32), res.col.astype(np.int32), res.data.astype(np.float32)


def main():
    out = {}
    m1, m2 = dup_inputs()
    for nm, m in (("dup_m1", m1), ("dup_m2", m2)):
        out[f"in/{nm}/data"], out[f"in/{nm}/indices"], out[f"in/{nm}/indptr"] = m.data, m.indices, m.indptr
        out[f"in/{nm}/shape"] = np.asarray(m.shape, dtype=np.int64)
    for name, fn, kw in (("dup_dot", "dot_product", {}), ("dup_cosine", "cosine", {}), ("dup_dot_thr", "dot_product", dict(threshold=0.5)),
                         ("dup_jaccard_shrink", "jaccard", dict(shrink=1.0))):
        res = getattr(ref, fn)(m1.copy(), m2.copy(), k=8, verbose=False, **kw)
        out[f"out/{name}/row"], out[f"out/{name}/col"], out[f"out/{name}/val"] = coo_triples(res)
        print(name, "stored", res.nnz, list(zip(res.row.tolist(), res.col.tolist(), np.round(res.data, 4).tolist()))[:12])

    u = p3_inputs()
    out["in/p3_m/data"], out["in/p3_m/indices"], out["in/p3_m/indptr"] = u.data, u.indices, u.indptr
    out["in/p3_m/shape"] = np.asarray(u.shape, dtype=np.int64)
    for name, fn, kw in (("p3_alpha4", "p3alpha", dict(alpha=4.0)), ("rp3_alpha4_beta", "rp3beta", dict(alpha=4.0, beta=0.3))):
        res = getattr(ref, fn)(u.copy(), k=12, verbose=False, **kw)
        out[f"out/{name}/row"], out[f"out/{name}/col"], out[f"out/{name}/val"] = coo_triples(res)
        print(name, "stored", res.nnz)

    np.savez_compressed(HERE / "quirks_golden.npz", **out)
    print("wrote", len(out), "arrays,", (HERE / "quirks_golden.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
