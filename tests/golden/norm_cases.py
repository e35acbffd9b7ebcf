"""Cases of the row-normaliser fixtures (shared by the generator and the tests): name -> (function, input, kwargs)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def build_inputs():
    rng = np.random.default_rng(42)
    A = sp.random_array((200, 100), density=0.05, format="csr", dtype=np.float32, random_state=rng)        # the reference tests' shape
    B = sp.random_array((60, 300), density=0.2, format="csr", dtype=np.float32, random_state=rng).tolil()
    B[5, :] = 0                                                                                              # empty rows, an empty column
    B[:, 7] = 0
    B = sp.csr_array(B.tocsr())
    B.data[::7] *= -1                                                                                        # negative entries (max / l1)
    C = sp.csr_array(sp.random_array((80, 50), density=0.3, format="csr", dtype=np.float64, random_state=rng))   # float64 stays float64
    D = sp.csr_array((rng.integers(1, 6, size=A.nnz).astype(np.int64), A.indices.copy(), A.indptr.copy()), shape=A.shape)   # counts (int -> float32)
    return {"A": A, "B": B, "C": C, "D": D}


def build_cases():
    cases = []
    for inp in ("A", "B", "C", "D"):
        for norm in ("l1", "l2", "max"):
            for axis in (1, 0):
                cases.append((f"{inp}_normalize_{norm}_ax{axis}", "normalize", inp, dict(norm=norm, axis=axis)))
    for inp in ("A", "C", "D"):
        cases.append((f"{inp}_tfidf_default", "tfidf", inp, {}))
        cases.append((f"{inp}_bm25_default", "bm25", inp, {}))
        cases.append((f"{inp}_bm25plus_default", "bm25plus", inp, {}))
    for tf in ("binary", "raw", "sqrt", "freq", "log"):
        for idf in ("unary", "base", "smooth", "prob", "bm25"):
            cases.append((f"A_tfidf_{tf}_{idf}", "tfidf", "A", dict(tf_mode=tf, idf_mode=idf, logbase=2.0)))
    for tf, idf in (("log", "smooth"), ("sqrt", "base"), ("binary", "bm25"), ("freq", "prob")):
        cases.append((f"D_bm25_{tf}_{idf}_ax0", "bm25", "D", dict(tf_mode=tf, idf_mode=idf, axis=0, k1=1.6, b=0.4)))
        cases.append((f"D_bm25plus_{tf}_{idf}", "bm25plus", "D", dict(tf_mode=tf, idf_mode=idf, k1=2.0, b=0.9, delta=0.5, logbase=10.0)))
    return cases
