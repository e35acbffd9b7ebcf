"""Generate tests/golden/norm_golden.npz by running the REFERENCE's normalisers (build container only; see make_golden.py
for the recipe that builds the reference into /tmp/simref_install):

    cd /tmp && PYTHONPATH=/tmp/simref_install:/root/repo python /root/repo/tests/golden/make_norm_golden.py

The fixture holds DATA only: the seeded input matrices and, per case of norm_cases.py, the CSR the reference returns."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

import norm_cases as NC  # noqa: E402

import similaripy as ref  # noqa: E402  (the reference)

assert "/root/repo" not in ref.__file__, "this script must import the REFERENCE package"


def main():
    inputs = NC.build_inputs()
    out = {}
    for name, m in inputs.items():
        out[f"in/{name}/data"], out[f"in/{name}/indices"], out[f"in/{name}/indptr"] = m.data, m.indices, m.indptr
        out[f"in/{name}/shape"] = np.array(m.shape)
    for name, fn, inp, kw in NC.build_cases():
        before = inputs[inp].copy()
        res = getattr(ref, fn)(inputs[inp], **kw)
        assert (before != inputs[inp]).nnz == 0, "inplace=False must not modify the input"
        res.sort_indices()
        out[f"out/{name}/data"], out[f"out/{name}/indices"], out[f"out/{name}/indptr"] = res.data, res.indices, res.indptr
    np.savez_compressed(HERE / "norm_golden.npz", **out)
    print(f"wrote {len(NC.build_cases())} cases, {sum(v.nbytes for v in out.values()) / 1e6:.2f} MB uncompressed")


if __name__ == "__main__":
    main()
