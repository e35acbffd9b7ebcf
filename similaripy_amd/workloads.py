"""Seeded synthetic inputs of the BASELINE.json configs (SURVEY §8d "canonical synthetic inputs").

Shared by bench.py, scripts/run_benchmarks.py and the full-size GPU tests so that every one of them runs on the
same matrices.  Nothing here touches the device.

  C1      sp.random_array((10_000, 20_000), density=0.01, float32, default_rng(0))            cosine k=50
  C2/C3   fixed-degree CSR 1M x 100k, 64 nnz/row, default_rng(12345)                          cosine / s_plus k=100
  C4      MovieLens-32M stand-in: 200 948 users x 84 432 items, nnz 32 000 204 exactly,
          Zipf item popularity, log-normal user activity, ratings in {0.5, ..., 5.0}           p3alpha / rp3beta k=200
  C5      urm = fixed-degree users x 100k items, W = cosine(urm_small.T, k=100)               dot_product + filter_cols
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

ML32M_USERS, ML32M_ITEMS, ML32M_NNZ = 200_948, 84_432, 32_000_204      # tests/benchmarks/README.md:194 of the reference


def c1_matrix(seed: int = 0) -> sp.csr_array:
    """BASELINE configs[0]: the reference tests' own generator (tests/test_similarity.py:284-286) at 10k x 20k."""
    return sp.random_array((10_000, 20_000), density=0.01, format="csr", dtype=np.float32,
                           random_state=np.random.default_rng(seed))


def fixed_degree_csr(n_rows: int, n_cols: int, nnz_row: int, seed: int) -> sp.csr_array:
    """SURVEY §8d canonical generator for C2/C3/C5: `nnz_row` uniform columns per row (duplicates merged), U[0,1) values."""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, n_cols, (n_rows, nnz_row), dtype=np.int32)
    cols.sort(axis=1)
    data = rng.random(n_rows * nnz_row, dtype=np.float32)
    indptr = np.arange(0, n_rows * nnz_row + 1, nnz_row, dtype=np.int32)   # int32 like the reference's kernel
    m = sp.csr_array((data, cols.ravel(), indptr), shape=(n_rows, n_cols))
    m.sum_duplicates()
    m.data[m.data == 0] = np.float32(0.5)   # rng.random can return exactly 0; keep nnz structural
    return m


def movielens_like_urm(n_users: int = ML32M_USERS, n_items: int = ML32M_ITEMS, nnz: int = ML32M_NNZ, seed: int = 0,
                       shuffle_items: bool = True) -> sp.csr_array:
    """Users x items rating matrix with EXACTLY `nnz` distinct (user, item) pairs: Zipf(0.9) item popularity,
    log-normal(0, 1) user activity, ratings uniform in {0.5, 1.0, ..., 5.0}.  Items are shuffled (a real catalogue is
    not sorted by popularity)."""
    rng = np.random.default_rng(seed)
    act = rng.lognormal(mean=0.0, sigma=1.0, size=n_users)
    act_cdf = np.cumsum(act / act.sum())
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.9
    pop_cdf = np.cumsum(pop / pop.sum())
    keys = np.zeros(0, dtype=np.int64)
    want = int(nnz)
    if want > n_users * n_items:
        raise ValueError("more entries than cells")
    while keys.shape[0] < want:
        draw = int((want - keys.shape[0]) * 1.15) + 1024
        u = np.minimum(np.searchsorted(act_cdf, rng.random(draw)), n_users - 1).astype(np.int64)
        i = np.minimum(np.searchsorted(pop_cdf, rng.random(draw)), n_items - 1).astype(np.int64)
        fresh = np.unique(u * n_items + i)
        if keys.shape[0]:
            fresh = fresh[~np.isin(fresh, keys, assume_unique=True)]
        if keys.shape[0] + fresh.shape[0] > want:
            fresh = rng.permutation(fresh)[: want - keys.shape[0]]       # (a random subset keeps the distribution)
        keys = np.sort(np.concatenate((keys, fresh)))
    u = (keys // n_items).astype(np.int32)
    i = (keys % n_items).astype(np.int32)
    if shuffle_items:
        i = rng.permutation(n_items).astype(np.int32)[i]
    r = (rng.integers(1, 11, size=want) * 0.5).astype(np.float32)
    m = sp.csr_array((r, (u, i)), shape=(n_users, n_items))
    m.sum_duplicates()
    m.sort_indices()
    assert m.nnz == want
    return m
