"""Golden-vector case list shared by make_golden.py (which runs the REFERENCE) and the tests
(which run similaripy_amd).  Inputs are tiny seeded matrices; they are also stored in the
fixture so the tests do not depend on RNG stability.

Matrices follow the reference's own test generator
(tests/test_similarity.py:284-286: sp.random_array(..., format='csr', dtype=float32,
random_state=default_rng(seed))) and parameters follow its check_similarity()
(tests/test_similarity.py:236-245).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def _rand(shape, density, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return sp.random_array(shape, density=density, format='csr', dtype=dtype, random_state=rng)


def build_inputs():
    """name -> scipy sparse matrix (or ndarray) used by the cases."""
    A = _rand((300, 200), 0.05, 42)
    B = _rand((60, 40), 0.2, 3)
    # B with some empty rows and an explicit stored zero.  The LAST row stays non-empty: with a
    # trailing empty row the reference itself raises IndexError in csr_sum's np.add.reduceat
    # (s_plus_utils.pyx:154) whenever l1 or l2 is active; similaripy_amd returns an empty output
    # row there instead (tests/test_host_logic.py covers it against the dense definition).
    E = B.copy().tolil()
    E[[0, 7, 30], :] = 0
    E = sp.csr_array(E.tocsr())
    E.data[3] = 0.0  # explicit zero: must be eliminated (s_plus.pyx:210-211)
    # signed data for negative-threshold behaviour
    S = A.copy()
    S.data = (S.data - 0.5).astype(np.float32)
    # recommender-style: urm (users x items) and a dense-ish item-item model
    URM = _rand((50, 80), 0.08, 7)
    W = _rand((80, 80), 0.6, 8)
    # integer and float64 inputs
    A_f64 = sp.csr_array(A, dtype=np.float64)
    A_int = sp.csr_array((np.ceil(A.data * 5).astype(np.int64), A.indices, A.indptr), shape=A.shape)
    M2 = _rand((200, 120), 0.06, 12)     # rectangular explicit matrix2 for A (300 x 200)
    pop1 = np.linspace(1.0, 4.0, 300).astype(np.float32)
    pop2 = np.linspace(2.0, 9.0, 300).astype(np.float32)
    return dict(A=A, B=B, E=E, S=S, URM=URM, W=W, A_f64=A_f64, A_int=A_int, M2=M2, pop1=pop1, pop2=pop2)


_ALL9 = [
    ("dot_product", {}),
    ("cosine", {}),
    ("asymmetric_cosine", dict(alpha=0.2)),
    ("jaccard", {}),
    ("dice", {}),
    ("tversky", dict(alpha=0.8, beta=0.4)),
    ("p3alpha", dict(alpha=0.8)),
    ("rp3beta", dict(alpha=0.8, beta=0.4)),
    ("s_plus", dict(l1=0.5, l2=0.5, l3=1, t1=1, t2=1, c1=0.5, c2=0.5, alpha=1, beta1=0, beta2=0,
                    pop1='none', pop2='sum')),
]


def build_cases():
    """List of dicts: name, fn, m1 (input key), optional m2 / filter_cols / target_cols given as
    input keys (prefix '@') or literals, kwargs."""
    cases = []

    def add(name, fn, m1, **kw):
        cases.append(dict(name=name, fn=fn, m1=m1, kw=kw))

    for fn, kw in _ALL9:
        add(f"A_{fn}_k10", fn, "A", k=10, **kw)
        add(f"B_{fn}_k5", fn, "B", k=5, **kw)
    for st in ("stabilized", "bayesian", "additive"):
        add(f"A_cosine_shrink10_{st}", "cosine", "A", k=10, shrink=10, shrink_type=st)
        add(f"B_tversky_shrink2_{st}", "tversky", "B", k=5, shrink=2, shrink_type=st, alpha=0.3, beta=0.6)
    add("A_cosine_binary", "cosine", "A", k=10, binary=True)
    add("A_jaccard_binary", "jaccard", "A", k=10, binary=True)
    add("A_rp3beta_binary", "rp3beta", "A", k=10, binary=True, alpha=0.8, beta=0.4)
    add("A_cosine_thr", "cosine", "A", k=10, threshold=0.05)
    add("A_dot_thr_high", "dot_product", "A", k=10, threshold=0.9)
    add("A_cosine_target_rows", "cosine", "A", k=10, target_rows=[7, 2, 150, 299])
    add("A_cosine_filter_array", "cosine", "A", k=10, filter_cols=list(range(0, 300, 3)))
    add("A_cosine_target_array", "cosine", "A", k=10, target_cols=[1, 2, 3, 4, 5, 6, 50, 51, 250, 1000, -4])
    add("A_cosine_target_and_filter_array", "cosine", "A", k=10,
        target_cols=[1, 2, 3, 4, 5, 6], filter_cols=[2, 3])
    add("URM_dot_filter_matrix", "dot_product", "URM", m2="@W", k=20, filter_cols="@URM")
    add("URM_dot_target_matrix", "dot_product", "URM", m2="@W", k=20, target_cols="@URM")
    add("URM_dot_filter_matrix_target_rows", "dot_product", "URM", m2="@W", k=80,
        filter_cols="@URM", target_rows=[1, 14, 8])
    add("URM_cos_filter_matrix_target_array", "cosine", "URM", m2="@W", k=15,
        filter_cols="@URM", target_cols=list(range(0, 80, 2)))
    add("R_cosine_rect_m2", "cosine", "A", m2="@M2", k=10)
    add("R_splus_rect_m2", "s_plus", "A", m2="@M2", k=10, l1=0.3, l2=0.6, l3=0.2, t1=0.7, t2=0.2,
        c1=0.4, c2=0.7, pop1="@pop1", pop2='sum', beta1=0.5, beta2=0.3, shrink=1.5)
    add("B_cosine_k_gt_ncols", "cosine", "B", k=500)
    add("B_dot_full_k", "dot_product", "B", k=60)
    add("A_splus_a1_noshrink", "s_plus", "A", k=10, l1=0, l2=0, l3=0, alpha=0.5)       # A.3 #10: raw dot
    add("A_splus_a1_shrink1", "s_plus", "A", k=10, l1=0, l2=0, l3=0, alpha=0.5, shrink=1)
    add("A_splus_a1_tversky", "s_plus", "A", k=10, l1=1, l2=0, l3=0, alpha=1.5, t1=0.5, t2=0.5)
    add("A_splus_pop_arrays", "s_plus", "A", k=10, l1=0.2, l2=0.2, l3=1, pop1="@pop1", pop2="@pop2",
        beta1=0.7, beta2=0.3)
    add("A_cosine_f64", "cosine", "A_f64", k=10)
    add("A_rp3beta_f64", "rp3beta", "A_f64", k=10, alpha=0.8, beta=0.4)
    add("A_cosine_int", "cosine", "A_int", k=10)
    add("A_p3alpha_int", "p3alpha", "A_int", k=10, alpha=0.8)
    add("E_cosine_empty_rows", "cosine", "E", k=5)
    add("E_rp3beta_empty_rows", "rp3beta", "E", k=5, alpha=0.8, beta=0.4)
    add("S_dot_signed_thr_neg", "dot_product", "S", k=10, threshold=-0.05)
    add("S_dot_signed_thr0", "dot_product", "S", k=10)
    add("A_cosine_block64", "cosine", "A", k=10, block_size=64)
    add("A_cosine_blocknone", "cosine", "A", k=10, block_size=None)
    add("A_cosine_k1", "cosine", "A", k=1)
    add("A_cosine_csr_out", "cosine", "A", k=10, format_output='csr')
    return cases


def resolve(value, inputs):
    if isinstance(value, str) and value.startswith("@"):
        return inputs[value[1:]]
    return value


def call_kwargs(case, inputs):
    """(positional matrix1, kwargs) for the public function of `case`."""
    kw = {}
    for key, val in case["kw"].items():
        if key == "m2":
            kw["matrix2"] = resolve(val, inputs)
        else:
            kw[key] = resolve(val, inputs)
    kw.setdefault("verbose", False)
    return inputs[case["m1"]], kw
