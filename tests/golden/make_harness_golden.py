"""Generates tests/golden/harness_golden.json by running the REFERENCE's benchmark harness functions on the harness test's small
ratings file (build container only: needs the reference built as SURVEY Appendix B describes).

    cd /tmp && PYTHONPATH=/tmp/simref_install:/root/reference/tests/benchmarks python /root/repo/tests/golden/make_harness_golden.py

What is recorded: for the default suite of run_benchmarks.py (dot_product, cosine, rp3beta; k = 10, shrink 0, threshold 0) the
`nnz` and `avg_neighbors` the reference's benchmark_similarity() reports (benchmark.py:88-215: item-item on URM.T, the wrappers'
default COO output), plus the URM's shape / nnz as the reference's ratings.csv loader builds it (dataset_loaders.py:94-116; the
loader itself downloads, so its mapping is restated here from the same DataFrame calls).  Data only: no reference source is copied."""
import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import harness_case as hc          # noqa: E402
import benchmark as ref_bench      # noqa: E402   (the reference's tests/benchmarks/benchmark.py)
import similaripy                  # noqa: E402   (the reference build)

assert "/tmp/simref_install" in similaripy.__file__, similaripy.__file__
u, i, r = hc.ratings_rows()
df = pd.DataFrame({"userId": u, "movieId": i, "rating": r})
umap = {x: n for n, x in enumerate(df["userId"].unique())}
imap = {x: n for n, x in enumerate(df["movieId"].unique())}
URM = sp.csr_array((df["rating"].values, (df["userId"].map(umap).values, df["movieId"].map(imap).values)),
                   shape=(len(umap), len(imap)), dtype=np.float32)
out = {"generator": "tests/golden/make_harness_golden.py", "reference_version": similaripy.__version__, "k": hc.K,
       "urm_shape": list(URM.shape), "urm_nnz": int(URM.nnz), "results": {}}
for name in ("dot_product", "cosine", "rp3beta"):
    res = ref_bench.benchmark_similarity(URM, similarity_type=name, k=hc.K, shrink=0, threshold=0, verbose=False)
    S = res["similarity_matrix"].tocsr()
    S.eliminate_zeros()
    out["results"][name] = {"nnz": int(res["nnz"]), "avg_neighbors": float(res["avg_neighbors"]), "n_items": int(res["n_items"]),
                            "nonzero_entries": int(S.nnz), "value_sum": float(np.float64(S.data.astype(np.float64).sum()))}
(HERE / "harness_golden.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
