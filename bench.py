#!/usr/bin/env python
"""Benchmark of the hot path: top-k sparse row similarity on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N == 1: this process.  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL).
  One JSON line on stdout from rank 0.

Metric (BASELINE.json): similarity rows/s (+ achieved algorithmic HBM GB/s), cosine k=100 on CSR.
Workload at N=1 = BASELINE.json configs[1] ("C2"): cosine, m1 = 1M x 100k fixed-degree 64 nnz/row
(SURVEY §8d canonical generator, seed 12345), m2 = m1.T, k = 100.  A "step" is one pass of the kernel
over all target rows with every operand already resident in HBM.

N > 1 ("weak"): every rank owns 1M target rows of its own (m1 shard seeded 12345+rank), m2 / Y* are the
replicated operands (rank 0's matrix transposed), no collective during compute, and the step ends
with the single RCCL gather of the (cols, values) slabs on rank 0 — inside the timed region.

Extra objects on the JSON line:
  roofline     — algorithmic bytes per launch (16*nnz1 + 8*MACs + 8*k per row) over the average
                 duration of the dominant kernel (sp_knn_sparse_kernel), measured with HIP events recorded on the
                 launch stream around that launch in K extra passes of the same step right after the timed region
                 (the timed steps themselves stay asynchronous).
  cpu_baseline — the reference kernel itself (oracle/_ref, kind "reference"; or the C port) timed on the
                 host cores of this box on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def fixed_degree_csr(n_rows: int, n_cols: int, nnz_row: int, seed: int) -> sp.csr_array:
    """SURVEY §8d canonical generator for C2/C3."""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, n_cols, (n_rows, nnz_row), dtype=np.int32)
    cols.sort(axis=1)
    data = rng.random(n_rows * nnz_row, dtype=np.float32)
    indptr = np.arange(0, n_rows * nnz_row + 1, nnz_row, dtype=np.int32)   # int32 like the reference's kernel
    m = sp.csr_array((data, cols.ravel(), indptr), shape=(n_rows, n_cols))
    m.sum_duplicates()
    m.data[m.data == 0] = np.float32(0.5)   # rng.random can return exactly 0; keep nnz structural
    return m


def algorithmic_bytes(call) -> tuple[int, int]:
    """(bytes, MACs) over call.targets — BASELINE.md §4."""
    nnz2 = np.diff(call.m2_indptr).astype(np.int64)
    per_entry = nnz2[call.m1_indices]
    csum = np.concatenate(([0], np.cumsum(per_entry)))
    macs_row = csum[call.m1_indptr[1:]] - csum[call.m1_indptr[:-1]]
    nnz1_row = np.diff(call.m1_indptr).astype(np.int64)
    t = call.targets
    macs = int(macs_row[t].sum())
    nbytes = int(16 * nnz1_row[t].sum() + 8 * macs + 8 * call.k * t.shape[0])
    return nbytes, macs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c1"])
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--nnz-row", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--table-slots", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--num-wgs", type=int, default=0)
    ap.add_argument("--load-pct", type=int, default=0)
    ap.add_argument("--static-sched", action="store_true")
    ap.add_argument("--no-sparse-path", action="store_true", help="force the generic windowed path (A/B)")
    ap.add_argument("--no-fold", action="store_true", help="do not fold the column term into the m2 stream (A/B)")
    ap.add_argument("--dbg", type=int, default=0, help="kernel ablation bits (profiling only; results invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from similaripy_amd import _abi, _host
    from similaripy_amd.device import DeviceProblem

    _abi.require_device()

    # ---------------- workload ----------------
    if args.workload in ("c2", "c3"):
        n_rows, n_cols, nnz_row, k = 1_000_000, 100_000, 64, 100
    else:  # c1: BASELINE configs[0], sps.random 10k x 20k d=0.01, k=50
        n_rows, n_cols, nnz_row, k = 10_000, 20_000, 200, 50
    n_rows = args.rows or n_rows
    n_cols = args.cols or n_cols
    nnz_row = args.nnz_row or nnz_row
    k = args.k or k
    kern_kw = dict(l2=1, c1=0.5, c2=0.5) if args.workload != "c3" else dict(l1=0.5, l2=0.5, stabilized_shrink=10.0)
    sim_name = "cosine" if args.workload != "c3" else "s_plus(l1=.5,l2=.5,shrink=10)"

    t0 = time.perf_counter()
    m0 = fixed_degree_csr(n_rows, n_cols, nnz_row, 12345)
    if rank == 0:
        m1, m2 = m0, None                     # m2 = m1.T
    else:
        m1, m2 = fixed_degree_csr(n_rows, n_cols, nnz_row, 12345 + rank), m0.T
    call = _host.prepare(m1, m2, k=k, **kern_kw)
    t_prep = time.perf_counter() - t0
    nbytes, macs = algorithmic_bytes(call)
    log(f"rank {rank}: {sim_name} {n_rows}x{n_cols} nnz/row~{m1.nnz / n_rows:.2f} k={k}: "
        f"MACs/row={macs / n_rows:.0f}, algorithmic {nbytes / 1e9:.1f} GB/launch, host prep {t_prep:.1f}s")

    prob = DeviceProblem(call, dev)
    cols, vals, counts, _ = prob.alloc_outputs()
    tuning = dict(table_slots=args.table_slots, threads_per_wg=args.threads, num_wgs=args.num_wgs, load_pct=args.load_pct, dbg=args.dbg,
                  no_sparse_path=args.no_sparse_path, no_fold=args.no_fold)

    gathered = None
    if world > 1 and rank == 0:
        gathered = ([torch.empty_like(cols) for _ in range(world)], [torch.empty_like(vals) for _ in range(world)])

    ev_pairs = []

    def step(timed: bool):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()                                    # torch's current stream == the launch stream
        prob.run(cols, vals, counts, static_sched=args.static_sched, **tuning)
        e1.record()
        if timed:
            ev_pairs.append((e0, e1))
        if world > 1:                                  # the one collective of the path: slabs -> rank 0
            dist.gather(cols, gathered[0] if rank == 0 else None, dst=0)
            dist.gather(vals, gathered[1] if rank == 0 else None, dst=0)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    step_ms = [a.elapsed_time(b) for a, b in ev_pairs]       # whole step on the launch stream: prep launches + row kernels
    # dominant kernel: hipEvents around its launch inside the library, K more passes of the same step (untimed region)
    infos = [prob.run(cols, vals, counts, time_kernel=True, static_sched=args.static_sched, phase_timers=False, **tuning) for _ in range(max(1, args.steps))]
    info = prob.run(cols, vals, counts, time_kernel=True, static_sched=args.static_sched, **tuning)     # one pass with the in-kernel phase timers
    sparse_ms = float(np.mean([i["sparse_kernel_ms"] for i in infos]))
    generic_ms = float(np.mean([i["generic_kernel_ms"] for i in infos]))
    dominant = "sp_knn_sparse_kernel" if sparse_ms >= generic_ms else "sp_knn_generic_kernel"
    kern_avg_s = max(sparse_ms, generic_ms) / 1e3
    n_kept = int(counts.sum().item())

    if rank != 0:
        dist.destroy_process_group()
        return

    total_rows = n_rows * world
    value = total_rows * args.steps / elapsed
    achieved = nbytes / kern_avg_s / 1e9
    traffic = None
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if tfile.exists():
        try:
            traffic = json.loads(tfile.read_text()).get(f"{args.workload}:{n_rows}x{n_cols}x{nnz_row}:k{k}", {}).get("bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "similarity rows/sec, cosine k=100 on CSR" if args.workload == "c2" else f"similarity rows/sec, {sim_name} k={k} on CSR",
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{sim_name} on fixed-degree CSR {n_rows}x{n_cols}, nnz/row={nnz_row}, k={k}, m2=m1.T "
                        f"(BASELINE configs[1]{' x ' + str(world) + ' ranks, own 1M-row m1 shard each, m2 replicated' if world > 1 else ''})",
            "rows_per_gpu": n_rows, "cols": n_cols, "nnz_per_row": nnz_row, "k": k,
            "macs_per_row": macs / n_rows,
            "parallelism": f"row-sharded x{world}" + (", gather to rank 0 in the step" if world > 1 else ""),
            "kept_entries": n_kept, "generic_windows_per_row": info["passes_total"] / n_rows,
            "phase_share": phase_share(info),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "kernel": dominant, "kernel_ms_avg": kern_avg_s * 1e3, "algorithmic_bytes_per_launch": nbytes,
            "step_ms_avg_on_stream": float(np.mean(step_ms)), "sparse_kernel_ms": sparse_ms, "generic_kernel_ms": generic_ms,
        },
    }

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(call, args.cpu_seconds)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def phase_share(info) -> dict:
    """Share of workgroup-lane-0 shader cycles per kernel phase (in-kernel s_memtime counters)."""
    # (names of include/sp_knn.h phase_cycles[0..8]; see there for what each covers in the two row kernels)
    names = ("setup", "segments", "accumulate", "drain", "select", "output", "sweep1", "sweep2", "csdrain")
    cyc = info.get("phase_cycles", [0] * 12)
    tot = float(sum(cyc[:9])) or 1.0
    d = {n: round(c / tot, 4) for n, c in zip(names, cyc[:9])}
    d["cycles_per_wg"] = tot / max(1, info.get("num_wgs", 1))
    d["rows_sparse_path"], d["generic_windows"] = cyc[9], cyc[11]
    d["rows_fallback_cs_full"], d["rows_fallback_u_overflow"] = cyc[10] & 0xFFFFFFFF, cyc[10] >> 32
    return d


def cpu_baseline(call, budget_s: float) -> dict:
    """Time the CPU kernel on a bounded sample (first S target rows) of the same workload."""
    from oracle import splus_oracle as so
    import copy

    kind = "reference" if so.available("reference") else "port"
    cores = so.max_threads(kind)

    def run(n, block_size):
        c = copy.copy(call)
        c.targets = np.ascontiguousarray(call.targets[:n])
        t0 = time.perf_counter()
        so.run_kernel(c, kind, num_threads=0, block_size=block_size)
        return time.perf_counter() - t0

    n_total = call.targets.shape[0]
    probe = min(n_total, 4000)
    run(min(n_total, 500), 0)                               # warm the thread team / page in
    rate = probe / run(probe, 0)
    sample = int(max(probe, min(n_total, rate * budget_s)))
    t = run(sample, 0)
    val = sample / t
    # the reference's default column blocking (block_size=0 -> 262144, s_plus.pyx:218-225); without its
    # popularity reorder, which only changes slot order — reported for completeness on a smaller sample
    s2 = max(probe, sample // 4)
    val_blocked = s2 / run(s2, 262144)
    log(f"cpu_baseline[{kind}] {cores} threads: unblocked {val:.0f} rows/s on {sample} rows; blocked(262144) {val_blocked:.0f} rows/s")
    return {
        "value": val, "unit": "rows/s", "cores": cores, "kind": kind,
        "sample": f"first {sample} of {n_total} target rows of the same workload, block_size off (the faster CPU variant), "
                  f"{t:.1f}s wall, all {cores} host threads (OpenMP dynamic schedule)",
        "blocked_262144_rows_per_s": val_blocked,
    }


if __name__ == "__main__":
    main()
