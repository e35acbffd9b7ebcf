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


def test_yambda_local_reader_maps_ids_to_ranks_and_sums_repeats(tmp_path):
    """dataset_loaders.py:136-232 of the reference, from a local copy of the hub layout: pd.Categorical codes (ranks of the
    sorted ids), implicit ones, repeated (user, item) events summed by the COO -> CSR conversion."""
    import pandas as pd
    rb = _load_module()
    d = tmp_path / "yambda" / "flat" / "50m"
    d.mkdir(parents=True)
    ev = [(900, 12, 1), (17, 40, 2), (900, 40, 3), (17, 40, 4), (230, 5, 5), (900, 12, 6)]
    pd.DataFrame(ev, columns=["uid", "item_id", "timestamp"]).to_parquet(d / "multi_event.parquet")
    URM = rb.load_yambda(tmp_path, "50m", "multi_event", verbose=False)
    u_cat, i_cat = pd.Categorical([e[0] for e in ev]), pd.Categorical([e[1] for e in ev])          # the reference's mapping
    want = np.zeros((len(u_cat.categories), len(i_cat.categories)), np.float32)
    for uc, ic in zip(u_cat.codes, i_cat.codes):
        want[uc, ic] += 1.0
    assert isinstance(URM, sp.csr_array) and URM.dtype == np.float32
    np.testing.assert_array_equal(URM.toarray(), want)
    # csv fallback, bad names
    (d / "likes.csv").write_text("uid,item_id,timestamp\n" + "\n".join(f"{u},{i},{t}" for u, i, t in ev) + "\n")
    np.testing.assert_array_equal(rb.load_yambda(tmp_path, "50m", "likes", verbose=False).toarray(), want)
    import pytest
    with pytest.raises(ValueError):
        rb.load_yambda(tmp_path, "5m", "likes", verbose=False)
    with pytest.raises(FileNotFoundError):
        rb.load_yambda(tmp_path, "500m", "likes", verbose=False)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_harness_end_to_end_on_a_small_ratings_file(tmp_path):
    """scripts/run_benchmarks.py as a user runs it (run_benchmarks.py:319-378 of the reference): a 2000 x 1500 ratings.csv, the
    default suite, two rounds — the report has the reference's keys, and its nnz / avg_neighbors are what direct calls give."""
    import json
    import subprocess
    import similaripy_amd as sim
    rng = np.random.default_rng(8)
    n_users, n_items = 2000, 1500
    m = sp.random_array((n_users, n_items), density=0.02, format="coo", dtype=np.float32, random_state=rng)
    d = tmp_path / "ml-tiny"
    d.mkdir()
    order = rng.permutation(m.nnz)
    ratings = np.ceil(m.data[order] * 10) / 2
    (d / "ratings.csv").write_text("userId,movieId,rating,timestamp\n" + "\n".join(f"{u + 1},{i + 1},{r},0" for u, i, r in zip(m.row[order], m.col[order], ratings)) + "\n")
    out = tmp_path / "out"
    cmd = [sys.executable, str(ROOT / "scripts" / "run_benchmarks.py"), "--dataset", "movielens", "--version", "tiny", "--data-dir", str(tmp_path),
           "--k", "10", "--rounds", "2", "--quiet", "--output-dir", str(out), "--note", "harness test"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    files = list(out.glob("benchmark_movielens_tiny_*.json"))
    assert len(files) == 1
    rep = json.loads(files[0].read_text())
    assert set(rep) == {"metadata", "config", "datasets", "results"}
    for key in ("similaripy_version", "numpy_version", "scipy_version", "python_version", "cpu_model", "cpu_count", "git_hash", "timestamp", "note"):
        assert key in rep["metadata"], key
    assert rep["config"]["similarities"] == ["dot_product", "cosine", "rp3beta"] and rep["config"]["k"] == 10 and rep["config"]["rounds"] == 2
    (dkey, ds), = rep["datasets"].items()
    rb = _load_module()
    URM = rb.load_movielens(tmp_path, "tiny", verbose=False)
    assert ds["shape"] == list(URM.shape) and ds["nnz"] == URM.nnz
    res = rep["results"][dkey]
    for name, fn in (("dot_product", sim.dot_product), ("cosine", sim.cosine), ("rp3beta", sim.rp3beta)):
        r = res[name]
        assert set(r) == {"computation_time", "std_time", "throughput", "nnz", "avg_neighbors", "rounds", "all_times"}
        assert r["rounds"] == 2 and len(r["all_times"]) == 2 and r["computation_time"] > 0 and r["throughput"] > 0
        direct = fn(URM.T, k=10, shrink=0, threshold=0, verbose=False)
        assert r["nnz"] == direct.nnz and abs(r["avg_neighbors"] - round(direct.nnz / URM.shape[1], 1)) < 1e-9
