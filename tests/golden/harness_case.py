"""The small ratings file of the benchmark-harness test (tests/test_harness.py) — shared with the script that generated the
golden report from the REFERENCE's own harness (make_harness_golden.py), so that both sides read the same rows."""
import numpy as np
import scipy.sparse as sp

N_USERS, N_ITEMS, K = 2000, 1500, 10


def ratings_rows():
    """(userId, movieId, rating) rows in file order: 1-based ids, shuffled, ratings in {0.5, ..., 5.0}."""
    rng = np.random.default_rng(8)
    m = sp.random_array((N_USERS, N_ITEMS), density=0.02, format="coo", dtype=np.float32, random_state=rng)
    order = rng.permutation(m.nnz)
    ratings = np.ceil(m.data[order] * 10) / 2
    return m.row[order] + 1, m.col[order] + 1, ratings


def write_ratings_csv(path):
    u, i, r = ratings_rows()
    with open(path, "w") as f:
        f.write("userId,movieId,rating,timestamp\n" + "\n".join(f"{a},{b},{c},0" for a, b, c in zip(u, i, r)) + "\n")
