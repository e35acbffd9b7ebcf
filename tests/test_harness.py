"""CPU tier: the benchmark harness (scripts/run_benchmarks.py) — the ratings.csv loader reproduces the
reference loader's id mapping (dataset_loaders.py:94-116), and the report carries the reference's JSON schema."""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent


def _load_module():
    spec = importlib.util.spec_from_file_location("run_benchmarks", ROOT / "scripts" / "run_benchmarks.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["run_benchmarks"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_ratings_csv_loader_maps_ids_in_order_of_appearance(tmp_path):
    rb = _load_module()
    d = tmp_path / "ml-tiny"
    d.mkdir()
    rows = [(50, 900, 4.0), (7, 30, 2.5), (50, 30, 1.0), (3, 77, 5.0), (7, 900, 0.5), (3, 30, 3.5)]
    (d / "ratings.csv").write_text("userId,movieId,rating,timestamp\n" + "\n".join(f"{u},{i},{r},0" for u, i, r in rows) + "\n")
    URM = rb.load_movielens(tmp_path, "tiny", verbose=False)
    # the reference: user_id_map = {id: idx for idx, id in enumerate(df['userId'].unique())} (order of first appearance)
    umap, imap = {}, {}
    for u, i, _ in rows:
        umap.setdefault(u, len(umap))
        imap.setdefault(i, len(imap))
    want = np.zeros((len(umap), len(imap)), np.float32)
    for u, i, r in rows:
        want[umap[u], imap[i]] = r
    assert isinstance(URM, sp.csr_array) and URM.dtype == np.float32
    np.testing.assert_array_equal(URM.toarray(), want)


def test_report_schema_keys():
    """run_benchmarks.py:367-375 of the reference: the keys compare_benchmarks.py reads."""
    src = (ROOT / "scripts" / "run_benchmarks.py").read_text()
    for key in ("computation_time", "std_time", "throughput", "nnz", "avg_neighbors", "rounds", "all_times",
                "metadata", "config", "datasets", "results", "similaripy_version", "cpu_model", "git_hash"):
        assert f'"{key}"' in src, key
