#!/usr/bin/env python
"""Benchmark harness with the reference's report schema.

Mirrors tests/benchmarks/run_benchmarks.py + benchmark.py + dataset_loaders.py of the reference:

  * what is timed (benchmark.py:161-189): `time.perf_counter()` around ONE public wrapper call on `URM.T`
    (item-item similarity), dataset loading excluded; throughput = items / seconds;
  * aggregation (run_benchmarks.py:152-187): mean and POPULATION standard deviation over `--rounds`;
  * JSON report (run_benchmarks.py:319-378): metadata / config / datasets / results with, per similarity,
    `computation_time, std_time, throughput, nnz, avg_neighbors, rounds, all_times` — the keys upstream's
    compare_benchmarks.py reads, so a report written here can be compared with one written by the reference;
  * CLI (run_benchmarks.py:429-470): same option names and defaults (dot_product cosine rp3beta, k=100, shrink=0,
    threads / block size accepted — they are CPU knobs the GPU path ignores).

Datasets: `--dataset movielens` reads `<data-dir>/ml-<version>/ratings.csv` exactly as the reference's loader does
(dataset_loaders.py:45-133: userId / movieId remapped in order of appearance, float32 CSR) when the file exists; there is
no network here, so without it `--dataset movielens-synthetic` (default when the file is missing) builds the seeded
MovieLens-32M-shaped matrix of similaripy_amd.workloads (200 948 x 84 432, nnz 32 000 204).  `--dataset yambda` reads the
Yandex Music Yambda interactions the reference pulls from the HuggingFace hub (dataset_loaders.py:136-232) from a LOCAL copy
of that repository's layout, `<data-dir>/yambda/flat/<version>/<event-type>.parquet` (or the same name with .csv): uid / item_id
mapped to their ranks (pandas Categorical codes: sorted ids), implicit 1.0 per event, repeated events summed by the COO -> CSR
conversion.  `--dataset c2` is the fixed-degree matrix of BASELINE configs[1] (its URM is the transpose, so that URM.T is the
1M x 100k matrix).
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import subprocess
import sys
import time
from datetime import datetime
from pathlib import Path

import numpy as np
import scipy
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import similaripy_amd as sim                      # noqa: E402
from similaripy_amd import _abi, workloads        # noqa: E402

SIMILARITIES = {
    "dot_product": sim.dot_product, "cosine": sim.cosine, "asymmetric_cosine": sim.asymmetric_cosine, "jaccard": sim.jaccard,
    "dice": sim.dice, "tversky": sim.tversky, "p3alpha": sim.p3alpha, "rp3beta": sim.rp3beta, "splus": sim.s_plus, "s_plus": sim.s_plus,
}


def get_system_info() -> dict:
    """benchmark.py:18-85 — plus the device the work runs on."""
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if "model name" in line:
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    git_hash = "unknown"
    try:
        r = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5, cwd=ROOT)
        if r.returncode == 0:
            git_hash = r.stdout.strip()
    except Exception:
        pass
    try:
        device = _abi.backend_info(0)
    except Exception as exc:           # noqa: BLE001
        device = f"unavailable ({exc})"
    return {
        "similaripy_version": f"similaripy_amd {sim.__version__}",
        "numpy_version": np.__version__, "scipy_version": scipy.__version__, "python_version": platform.python_version(),
        "system": platform.system(), "arch": platform.machine(), "cpu_model": cpu_model, "cpu_count": os.cpu_count() or "unknown",
        "git_hash": git_hash, "timestamp": datetime.now().strftime("%Y-%m-%d %H:%M:%S"), "device": device,
    }


def load_movielens(data_dir: Path, version: str, verbose: bool):
    """dataset_loaders.py:45-133 without the download: ratings.csv -> float32 CSR, ids remapped in order of appearance."""
    import pandas as pd
    f = data_dir / f"ml-{version}" / "ratings.csv"
    if not f.exists():
        raise FileNotFoundError(f"Ratings file not found at {f} (no network here: place the extracted MovieLens archive there)")
    df = pd.read_csv(f)
    users, uidx = np.unique(df["userId"].values, return_index=True)
    items, iidx = np.unique(df["movieId"].values, return_index=True)
    # order of first appearance, as dict(enumerate(unique())) gives it upstream
    umap = np.empty(users.shape[0], np.int64); umap[np.argsort(uidx)] = np.arange(users.shape[0])
    imap = np.empty(items.shape[0], np.int64); imap[np.argsort(iidx)] = np.arange(items.shape[0])
    u = umap[np.searchsorted(users, df["userId"].values)]
    i = imap[np.searchsorted(items, df["movieId"].values)]
    URM = sp.csr_array((df["rating"].values, (u, i)), shape=(users.shape[0], items.shape[0]), dtype=np.float32)
    if verbose:
        print(f"Loaded {len(df)} ratings: URM {URM.shape}, nnz {URM.nnz}")
    return URM


YAMBDA_VERSIONS = ("50m", "500m")                      # dataset_loaders.py:33-42
YAMBDA_EVENTS = ("likes", "listens", "multi_event")     # dataset_loaders.py:146-147


def load_yambda(data_dir: Path, version: str = "50m", event_type: str = "multi_event", verbose: bool = True):
    """dataset_loaders.py:136-232 without the hub: the `flat/<version>/<event_type>.parquet` file of a local copy of
    yandex/yambda -> users x items CSR of float32 event counts, ids replaced by their ranks (what pd.Categorical(...).codes are)."""
    import pandas as pd
    if version not in YAMBDA_VERSIONS:
        raise ValueError(f"Unknown Yambda version '{version}'. Available: {list(YAMBDA_VERSIONS)}")
    if event_type not in YAMBDA_EVENTS:
        raise ValueError(f"Unknown Yambda event type '{event_type}'. Available: {list(YAMBDA_EVENTS)}")
    base = data_dir / "yambda" / "flat" / version / event_type
    if base.with_suffix(".parquet").exists():
        df = pd.read_parquet(base.with_suffix(".parquet"), columns=["uid", "item_id"])
    elif base.with_suffix(".csv").exists():
        df = pd.read_csv(base.with_suffix(".csv"), usecols=["uid", "item_id"])
    else:
        raise FileNotFoundError(f"Yambda file not found at {base}.parquet (no network here: place a copy of the hub repository's flat/ tree there)")
    uid = df["uid"].to_numpy(dtype=np.int64)
    iid = df["item_id"].to_numpy(dtype=np.int64)
    users, u = np.unique(uid, return_inverse=True)       # rank of every id among the sorted distinct ids = Categorical codes
    items, i = np.unique(iid, return_inverse=True)
    URM = sp.coo_array((np.ones(uid.shape[0], dtype=np.float32), (u, i)), shape=(users.shape[0], items.shape[0])).tocsr()
    if verbose:
        print(f"Loaded {uid.shape[0]} interactions: URM {URM.shape}, nnz {URM.nnz}")
    return URM


def load_URM(dataset: str, version: str, data_dir: Path, verbose: bool, event_type: str = "multi_event"):
    if dataset == "yambda":
        return load_yambda(data_dir, version, event_type, verbose), f"{version}-{event_type}"
    if dataset == "movielens":
        try:
            return load_movielens(data_dir, version, verbose), version
        except FileNotFoundError as exc:
            if verbose:
                print(f"{exc}\n-> using the seeded synthetic matrix of the MovieLens-32M shape instead")
            dataset = "movielens-synthetic"
    if dataset == "movielens-synthetic":
        return workloads.movielens_like_urm(), "32m-shape-seed0"
    if dataset == "c2":
        return workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345).T.tocsr(), "1Mx100k-64"
    raise ValueError(f"unknown dataset {dataset}")


def benchmark_similarity(URM, similarity_type="cosine", k=100, shrink=0, threshold=0, num_threads=0, verbose=True, **kw):
    """benchmark.py:88-214: item-item similarity on URM.T, wall clock of the public call."""
    item_matrix = URM.T
    fn = SIMILARITIES[similarity_type]
    block_size = kw.pop("block_size", 0)
    t0 = time.perf_counter()
    S = fn(item_matrix, k=k, shrink=shrink, threshold=threshold, verbose=verbose, num_threads=num_threads, block_size=block_size, **kw)
    dt = time.perf_counter() - t0
    n_items, nnz = S.shape[0], S.nnz
    assert S.shape[0] == S.shape[1] and nnz > 0
    return {"similarity_matrix": S, "computation_time": dt, "n_items": n_items, "nnz": nnz, "density": nnz / (n_items * n_items),
            "avg_neighbors": nnz / n_items, "throughput": n_items / dt, "similarity_type": similarity_type, "k": k, "shrink": shrink,
            "threshold": threshold, "block_size": block_size}


def main():
    ap = argparse.ArgumentParser(description="Benchmark suite of similaripy_amd with the reference's report schema")
    ap.add_argument("--dataset", default="movielens", choices=["movielens", "movielens-synthetic", "yambda", "c2"])
    ap.add_argument("--event-type", type=str, default="multi_event", choices=list(YAMBDA_EVENTS), help="Yambda event file")
    ap.add_argument("--version", type=str, default="32m")
    ap.add_argument("--data-dir", type=str, default="datasets")
    ap.add_argument("--similarities", nargs="+", default=["dot_product", "cosine", "rp3beta"])
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--shrink", type=float, default=0)
    ap.add_argument("--threshold", type=float, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--block-size", type=str, default="0")
    ap.add_argument("--format-output", default="coo", choices=["coo", "csr"], help="(the reference's harness uses the wrappers' default, coo)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed calls before the rounds (library load, device buffer cache); the reference has none")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--output-dir", type=str, default="bench_results")
    ap.add_argument("--note", type=str, default=None)
    args = ap.parse_args()
    verbose = not args.quiet
    block_size = None if args.block_size.lower() == "none" else int(args.block_size)

    sys_info = get_system_info()
    URM, version = load_URM(args.dataset, args.version, Path(args.data_dir), verbose, args.event_type)
    key = f"{args.dataset}:{version}"
    density = URM.nnz / (URM.shape[0] * URM.shape[1])
    if verbose:
        print(f"URM shape: {URM.shape}\nURM nnz: {URM.nnz:,}\nURM density: {density:.6%}")
    results = {}
    for s in args.similarities:
        kw = dict(format_output=args.format_output)
        for _ in range(args.warmup):
            benchmark_similarity(URM, s, args.k, args.shrink, args.threshold, args.threads, False, block_size=block_size, **kw)
        rounds = [benchmark_similarity(URM, s, args.k, args.shrink, args.threshold, args.threads, False, block_size=block_size, **kw) for _ in range(args.rounds)]
        times = [r["computation_time"] for r in rounds]
        avg = sum(times) / len(times)
        std = (sum((t - avg) ** 2 for t in times) / len(times)) ** 0.5 if len(times) > 1 else 0.0
        results[s] = {"computation_time": round(avg, 4), "std_time": round(std, 4), "throughput": round(sum(r["throughput"] for r in rounds) / len(rounds), 1),
                      "nnz": int(rounds[0]["nnz"]), "avg_neighbors": round(rounds[0]["avg_neighbors"], 1), "rounds": args.rounds, "all_times": [round(t, 4) for t in times]}
        if verbose:
            print(f"{s:>18}: {avg:.4f} +- {std:.4f} s   {results[s]['throughput']:.1f} items/s   nnz {results[s]['nnz']:,}")
    report = {
        "metadata": {**sys_info, "note": args.note or ""},
        "config": {"datasets": [[args.dataset, version]], "similarities": args.similarities, "k": args.k, "shrink": args.shrink, "threshold": args.threshold,
                   "num_threads": args.threads, "block_size": "none" if block_size is None else ("auto" if block_size == 0 else str(block_size)), "rounds": args.rounds},
        "datasets": {key: {"shape": list(URM.shape), "nnz": int(URM.nnz), "density": density}},
        "results": {key: results},
    }
    out_dir = Path(args.output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    out = out_dir / f"benchmark_{args.dataset}_{version}_{datetime.now().strftime('%Y%m%d_%H%M%S')}.json"
    out.write_text(json.dumps(report, indent=2))
    print(f"report written to {out}")


if __name__ == "__main__":
    main()
