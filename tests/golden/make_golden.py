"""Generate tests/golden/splus_golden.npz by running the REFERENCE implementation.

Runs only in the build container, where /root/reference can be built and imported
(SURVEY Appendix B):

    rm -rf /tmp/simref /tmp/simref_install && cp -r /root/reference /tmp/simref
    cd /tmp/simref && pip install --no-build-isolation --no-deps --no-index --target /tmp/simref_install .
    cd /tmp && PYTHONPATH=/tmp/simref_install:/root/repo python /root/repo/tests/golden/make_golden.py

The fixture holds DATA only: the seeded input matrices, and for every case in cases.py the
reference's output canonicalised per slot (real entries sorted by column, padding counted),
plus a few intermediates of the reference's host helpers (norm vectors, selectors) for the
host-logic tests.  No reference source text is stored.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import cases as C  # noqa: E402

import similaripy as ref  # noqa: E402  (the reference, from /tmp/simref_install)
from similaripy.cython_code import s_plus_utils as ref_utils  # noqa: E402
from similaripy.normalization import normalize as ref_normalize  # noqa: E402

assert "/root/repo" not in ref.__file__, "this script must import the REFERENCE package"


def canon_coo(res, targets, k):
    """COO result (n_targets*k stored triples incl. padding) -> counts, cols, vals (per slot, by column)."""
    rows, cols, vals = res.row.astype(np.int32), res.col.astype(np.int32), res.data.astype(np.float32)
    n = targets.shape[0]
    assert rows.shape[0] == n * k, (rows.shape, n, k)
    r, c, v = rows.reshape(n, k), cols.reshape(n, k), vals.reshape(n, k)
    real = r == targets[:, None]
    real &= ~((targets == 0)[:, None] & (c == 0) & (v == 0))
    counts = real.sum(axis=1).astype(np.int32)
    oc, ov = [], []
    for i in range(n):
        # padding must be the tail of the slot
        assert real[i, : counts[i]].all(), "padding is not at the tail of the slot"
        ci, vi = c[i, : counts[i]], v[i, : counts[i]]
        o = np.argsort(ci, kind="stable")
        oc.append(ci[o])
        ov.append(vi[o])
    return counts, np.concatenate(oc) if oc else np.zeros(0, np.int32), np.concatenate(ov) if ov else np.zeros(0, np.float32)


def main():
    inputs = C.build_inputs()
    out = {}
    manifest = {"numpy": np.__version__, "scipy": __import__("scipy").__version__,
                "reference_version": getattr(ref, "__version__", "?"), "cases": []}

    for name, m in inputs.items():
        if sp.issparse(m):
            m = m.tocsr()
            out[f"in/{name}/data"], out[f"in/{name}/indices"], out[f"in/{name}/indptr"] = m.data, m.indices, m.indptr
            out[f"in/{name}/shape"] = np.asarray(m.shape, dtype=np.int64)
        else:
            out[f"in/{name}/array"] = np.asarray(m)

    for case in C.build_cases():
        m1, kw = C.call_kwargs(case, inputs)
        fn = getattr(ref, case["fn"])
        # the reference mutates CSR inputs in place (eliminate_zeros): hand it copies
        kw2 = {a: (b.copy() if sp.issparse(b) else b) for a, b in kw.items()}
        res = fn(m1.copy(), **kw2)
        name = case["name"]
        n_rows = m1.shape[0]
        targets = np.asarray(kw.get("target_rows", np.arange(n_rows)), dtype=np.int32)
        m2 = kw.get("matrix2")
        n_cols = m2.shape[1] if m2 is not None else m1.shape[0]
        k_eff = min(int(kw.get("k", 100)), n_cols)
        entry = {"name": name, "k_eff": k_eff, "shape": list(res.shape), "format": kw.get("format_output", "coo")}
        if entry["format"] == "coo":
            assert isinstance(res, sp.coo_array)
            counts, cols, vals = canon_coo(res, targets, k_eff)
            out[f"out/{name}/counts"], out[f"out/{name}/cols"], out[f"out/{name}/vals"] = counts, cols, vals
            entry["stored_nnz"] = int(res.nnz)
        else:
            assert isinstance(res, sp.csr_array)
            r = res.copy()
            entry["has_sorted_indices_before_sort"] = bool(r.has_sorted_indices)
            r.sort_indices()
            out[f"out/{name}/indptr"] = r.indptr.astype(np.int64)
            out[f"out/{name}/cols"], out[f"out/{name}/vals"] = r.indices.astype(np.int32), r.data.astype(np.float32)
            entry["stored_nnz"] = int(res.nnz)
        assert res.dtype == np.float32
        manifest["cases"].append(entry)
        print(f"{name:45s} nnz={entry['stored_nnz']}")

    # ---- intermediates of the reference's host helpers (plain defs) ----
    A = inputs["A"].copy()
    A.eliminate_zeros()
    AT = sp.csr_array(A.T.tocsr())
    sq1, sq2 = ref_utils._build_squared_norms(A, AT)
    out["mid/A/sq1"], out["mid/A/sq2"] = np.asarray(sq1), np.asarray(sq2)
    xc, yc = ref_utils._build_cosine_normalization(np.asarray(sq1), np.asarray(sq2), 0.2, 0.8, 10.0)
    out["mid/A/xcos_c0.2_add10"], out["mid/A/ycos_c0.8_add10"] = np.asarray(xc), np.asarray(yc)
    xd, yd = ref_utils._build_depop_normalization(A, AT, 'sum', inputs["pop2"], 0.7, 0.3)
    out["mid/A/xdep_sum_p0.7"], out["mid/A/ydep_pop2_p0.3"] = np.asarray(xd), np.asarray(yd)
    out["mid/A/rowsum"] = np.asarray(ref_utils.csr_sum(A, 1))
    out["mid/A/colsum"] = np.asarray(ref_utils.csr_sum(A, 0))
    tc = ref_utils._compute_target_columns([2, 3, 400, -1], [1, 2, 3, 4, 5, 6, 299, 300], 300)
    out["mid/target_columns"] = np.asarray(tc)
    fd, fi, fp = ref_utils._filter_matrix_columns(AT, np.asarray(tc))
    out["mid/AT_filtered/data"], out["mid/AT_filtered/indices"], out["mid/AT_filtered/indptr"] = fd, fi, fp
    mode, ip, ix = ref_utils._build_column_selector(inputs["URM"])
    out["mid/URM_selector/indptr"], out["mid/URM_selector/indices"] = np.asarray(ip), np.asarray(ix)
    manifest["URM_selector_mode"] = int(mode)
    for nm in ("l1", "l2", "max"):
        for ax in (0, 1):
            r = ref_normalize(inputs["E"], norm=nm, axis=ax)
            r.sort_indices()
            out[f"mid/E_norm_{nm}_ax{ax}/data"], out[f"mid/E_norm_{nm}_ax{ax}/indices"], out[f"mid/E_norm_{nm}_ax{ax}/indptr"] = \
                r.data, r.indices, r.indptr

    np.savez_compressed(HERE / "splus_golden.npz", **out)
    (HERE / "splus_golden_manifest.json").write_text(json.dumps(manifest, indent=1))
    size = (HERE / "splus_golden.npz").stat().st_size
    print(f"wrote {len(out)} arrays, {size/1024:.0f} KiB")


if __name__ == "__main__":
    main()
