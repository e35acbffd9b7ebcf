#!/usr/bin/env python
"""Benchmark of the hot path: top-k sparse row similarity on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N == 1: this process.  N > 1: launched by torch.distributed.run, one rank per GPU (RCCL).
  One JSON line on stdout from rank 0.

Metric (BASELINE.json): similarity rows/s (+ achieved algorithmic HBM GB/s), cosine k=100 on CSR.
Workload at N=1 = BASELINE.json configs[1] ("C2"): cosine, m1 = 1M x 100k fixed-degree 64 nnz/row
(SURVEY §8d canonical generator, seed 12345), m2 = m1.T, k = 100.  A "step" is one pass of the kernel
over all target rows with every operand already resident in HBM.
--workload c1 | c3 | c4 | c5 time the other BASELINE configs the same way (not the headline line): c4 = rp3beta(alpha .8,
beta .4) item-item on the MovieLens-32M-shaped URM, k = 200 (p3alpha timed beside it); c5 = dot_product(urm, W.T,
filter_cols=urm), 1M users x 100k items per GPU, W = cosine top-100 of a 200k-user sample.

N > 1 goes through the shipped multi-GPU driver, `similaripy_amd.distributed.ShardedDeviceProblem`: every rank builds
the same problem, `partition_targets` cuts the target list into N contiguous work-balanced slices, each rank's slice is
resident on its own GPU (m2 / Y* replicated), no collective during compute, and the step ends with the single RCCL
gather of the (cols, values, counts) slabs on rank 0 — inside the timed region.
  --scaling weak (default): m1 = the 1M x 100k matrix of C2 stacked N times (N*1M rows), m2 = the transpose of ONE copy:
      N*1M output rows, exactly 1M per rank, every rank does the work of the N=1 run (at N=1 it IS the N=1 run).
  --scaling strong: the 1M rows of C2, cut N ways.

Extra objects on the JSON line:
  roofline     — algorithmic bytes per launch (16*nnz1 + 8*MACs + 8*k per row) over the average
                 duration of the dominant kernel (sp_knn_sparse_kernel), measured with HIP events recorded on the
                 launch stream around that launch in K extra passes of the same step right after the timed region
                 (the timed steps themselves stay asynchronous).
  cpu_baseline — the reference kernel itself (oracle/_ref, kind "reference"; or the C port) timed on the physical
                 host cores of this box (OMP_PROC_BIND=spread, OMP_PLACES=cores, in a child process so that the OpenMP
                 runtime sees them) on a bounded sample of the same workload: warm-up + 3 rounds, mean +- std, for the
                 reference's default column blocking (block_size=0 -> 262144: the headline `value`) and for blocking off
                 (rank 0, N=1 only).
  end_to_end_s — wall clock of the public call `similaripy_amd.cosine(m, k=100, format_output="csr")` (the reference
                 harness's definition, benchmark.py:168-189), host preprocessing and output assembly included.
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


from similaripy_amd.workloads import c1_matrix, fixed_degree_csr      # noqa: E402  (SURVEY §8d canonical generators)


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


def lib_source_sha() -> str:
    """Short hash of the kernel sources: profiles/hbm_traffic.json entries are valid for one build only."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "similaripy_amd" / "csrc").glob("*")):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c1", "c4", "c5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
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
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=3.0, help="target wall time of ONE cpu_baseline round")
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
    from similaripy_amd.distributed import ShardedDeviceProblem

    _abi.require_device()

    # ---------------- workload ----------------
    t0 = time.perf_counter()
    reps = world if args.scaling == "weak" else 1
    extra_cfg = {}
    public_call = None
    if args.workload in ("c1", "c2", "c3"):
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
        if args.workload == "c1" and not (args.rows or args.cols or args.nnz_row):
            m1 = c1_matrix()                                     # configs[0] exactly: sps.random
            gen = "sps.random d=0.01"
        else:
            m1 = fixed_degree_csr(n_rows, n_cols, nnz_row, 12345)
            gen = f"fixed-degree CSR nnz/row={nnz_row}"
        # the same problem on every rank (seeded); weak scaling: the matrix stacked `world` times over one copy's transpose
        m2 = m1.T.tocsr()

        def make_call(r):
            return _host.prepare(sp.vstack([m1] * r, format="csr") if r > 1 else m1, m2, k=k, **kern_kw)
        workload_txt = (f"{sim_name} on {gen} {n_rows}x{n_cols}, k={k}, m2=m1.T (BASELINE configs[{ {'c1': 0, 'c2': 1, 'c3': 2}[args.workload] }])"
                        + (f"; weak scaling: m1 = that matrix stacked {reps} times ({n_rows * reps} rows, {n_rows} per rank), m2 = one copy's transpose" if reps > 1 else ""))
        if args.workload == "c3":
            public_call = ("similaripy_amd.s_plus(m, l1=.5, l2=.5, shrink=10, k=%d, format_output='csr')" % k,
                           lambda sim: sim.s_plus(m1, l1=0.5, l2=0.5, shrink=10, k=k, verbose=False, format_output="csr"))
        else:
            public_call = ("similaripy_amd.cosine(m, k=%d, format_output='csr')" % k, lambda sim: sim.cosine(m1, k=k, verbose=False, format_output="csr"))
    elif args.workload == "c4":
        # BASELINE configs[3]: p3alpha + rp3beta, item-item on the MovieLens-32M URM (its seeded stand-in: no network), k = 200.
        # The step is the rp3beta call (p3alpha's stream + the popularity term); p3alpha is timed beside it (config.p3alpha_ms).
        from similaripy_amd.normalization import normalize
        from similaripy_amd.workloads import movielens_like_urm
        k, alpha, beta = args.k or 200, 0.8, 0.4
        urm = movielens_like_urm()
        m1 = urm.T.tocsr()
        n_rows, n_cols, nnz_row = m1.shape[0], m1.shape[1], int(round(m1.nnz / m1.shape[0]))
        pop_m2 = np.asarray(m1.T.sum(axis=0)).ravel()               # similarity.py:479 — BEFORE normalisation
        a_ = normalize(m1, norm="l1", axis=1); a_.data = np.power(a_.data, np.float32(alpha))
        b_ = normalize(m1.T.tocsr(), norm="l1", axis=1); b_.data = np.power(b_.data, np.float32(alpha))
        sim_name = f"rp3beta(alpha={alpha}, beta={beta})"

        def make_call(r):
            if r > 1:
                raise SystemExit("c4 has no weak-scaling form (one catalogue): use --scaling strong")
            return _host.prepare(a_, b_, k=k, weight_depop_matrix2=pop_m2, p2=beta, l3=1)
        extra_cfg["_p3alpha_call"] = lambda: _host.prepare(a_, b_, k=k)
        workload_txt = (f"{sim_name} item-item on the MovieLens-32M-shaped URM {urm.shape[0]}x{urm.shape[1]} nnz {urm.nnz} "
                        f"(workloads.movielens_like_urm, seed 0), k={k} (BASELINE configs[3]; p3alpha timed beside it)")
        public_call = (f"similaripy_amd.rp3beta(URM.T, alpha={alpha}, beta={beta}, k={k}, format_output='csr')",
                       lambda sim: sim.rp3beta(m1, alpha=alpha, beta=beta, k=k, verbose=False, format_output="csr"))
        if args.scaling == "weak" and world > 1:
            raise SystemExit("c4 has no weak-scaling form (one catalogue): use --scaling strong")
        reps = 1
    else:
        # BASELINE configs[4]: dot_product(urm, W.T, k=100, filter_cols=urm) — user scoring with the seen items excluded; one GPU's
        # share of the 10M-user job: 1M users x 100k items, 64 per user; W = cosine top-100 item model of a 200k-user sample
        import similaripy_amd as sim_pkg
        n_rows, n_cols, nnz_row, k = args.rows or 1_000_000, args.cols or 100_000, args.nnz_row or 64, args.k or 100
        urm = fixed_degree_csr(n_rows, n_cols, nnz_row, 12345)
        W = sim_pkg.cosine(urm[: min(n_rows, 200_000)].T.tocsr(), k=100, verbose=False, format_output="csr")
        Wt = W.T.tocsr()
        sim_name = "dot_product(urm, W.T, filter_cols=urm)"

        def make_call(r):
            u = sp.vstack([urm] * r, format="csr") if r > 1 else urm
            return _host.prepare(u, Wt, k=k, filter_cols=u)
        workload_txt = (f"{sim_name} on fixed-degree URM {n_rows}x{n_cols} nnz/row={nnz_row}, W = cosine top-100 of a 200k-user sample "
                        f"(nnz {W.nnz}), k={k} (BASELINE configs[4], one GPU's share)"
                        + (f"; weak scaling: the URM stacked {reps} times" if reps > 1 else ""))
        public_call = (f"similaripy_amd.dot_product(urm, W.T, k={k}, filter_cols=urm, format_output='csr')",
                       lambda sim: sim.dot_product(urm, Wt, k=k, filter_cols=urm, verbose=False, format_output="csr"))
    call = make_call(reps)
    t_prep = time.perf_counter() - t0
    nbytes, macs = algorithmic_bytes(call)                   # over ALL target slots of the job
    total_rows = call.n_targets
    log(f"rank {rank}: {sim_name} {n_rows}x{n_cols} nnz/row~{nnz_row} k={k}, {total_rows} target slots: "
        f"MACs/row={macs / total_rows:.0f}, algorithmic {nbytes / 1e9:.1f} GB/step, host prep {t_prep:.1f}s")

    tuning = dict(table_slots=args.table_slots, threads_per_wg=args.threads, num_wgs=args.num_wgs, load_pct=args.load_pct, dbg=args.dbg,
                  no_sparse_path=args.no_sparse_path, no_fold=args.no_fold)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(the_call, steps, warmup, detail):
        """K timed steps of the sharded problem (barrier + synchronize on both sides, max over ranks), then the per-kernel
        and per-rank figures from K more passes outside the timed region."""
        shard = ShardedDeviceProblem(the_call, device=dev)           # partition_targets + DeviceProblem of this rank's slice
        ev_pairs = []

        def step(timed: bool):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()                                    # torch's current stream == the launch stream
            shard.run(gather=False, static_sched=args.static_sched, **tuning)
            e1.record()
            if timed:
                ev_pairs.append((e0, e1))
            if world > 1:                                  # the ONE collective of the path: slabs -> rank 0
                shard.gather()

        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        res = {"elapsed": elapsed, "step_ms": [a.elapsed_time(b) for a, b in ev_pairs], "shard": shard}
        if not detail:
            return res
        # dominant kernel: hipEvents around its launch inside the library, K more passes of the same step (untimed region)
        infos = [shard.run(gather=False, time_kernel=True, static_sched=args.static_sched, phase_timers=False, **tuning) for _ in range(max(1, steps))]
        res["info"] = shard.run(gather=False, time_kernel=True, static_sched=args.static_sched, **tuning)     # one pass with the in-kernel phase timers
        res["sparse_ms"] = float(np.mean([i["sparse_kernel_ms"] for i in infos]))
        res["generic_ms"] = float(np.mean([i["generic_kernel_ms"] for i in infos]))
        res["call_ms"] = float(np.mean([i["kernel_ms"] for i in infos]))
        gather_ms = 0.0
        if world > 1:                                      # the gather alone, on the stream it runs on
            g = []
            for _ in range(max(1, steps)):
                fence()
                t1 = time.perf_counter()
                shard.gather()
                torch.cuda.synchronize()
                g.append((time.perf_counter() - t1) * 1e3)
            gather_ms = float(np.mean(g))
        local_bytes, local_macs = algorithmic_bytes(shard.prob.call)
        mine = {"rank": rank, "rows": int(shard.n_loc), "macs": int(local_macs), "algorithmic_bytes": int(local_bytes), "call_ms": res["call_ms"],
                "sparse_kernel_ms": res["sparse_ms"], "generic_kernel_ms": res["generic_ms"], "gather_ms": gather_ms}
        if world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            res["per_rank"] = allr
        else:
            res["per_rank"] = [mine]
        res["local_bytes"] = local_bytes
        return res

    main_res = measure(call, args.steps, args.warmup, True)
    shard, info = main_res["shard"], main_res["info"]
    elapsed, step_ms = main_res["elapsed"], main_res["step_ms"]
    sparse_ms, generic_ms = main_res["sparse_ms"], main_res["generic_ms"]
    dominant = "sp_knn_sparse_kernel" if sparse_ms >= generic_ms else "sp_knn_generic_kernel"
    kern_avg_s = max(sparse_ms, generic_ms) / 1e3
    n_kept = int(shard.pad_cnt.sum().item())
    local_bytes = main_res["local_bytes"]
    # the other scaling mode of the same workload, so that one multi-GPU run answers both questions: weak (per-GPU work fixed;
    # the headline line keeps the contract's default) and strong (the N = 1 job cut N ways: what ">= 6x at 8 GPUs" is about)
    other = None
    if world > 1 and args.workload in ("c1", "c2", "c3", "c5"):
        other_mode = "strong" if args.scaling == "weak" else "weak"
        del shard
        main_res["shard"] = None
        torch.cuda.empty_cache()
        oc = make_call(1 if other_mode == "strong" else world)
        orows = oc.n_targets
        ores = measure(oc, args.steps, args.warmup, True)
        other = {"scaling": other_mode, "value": orows * args.steps / ores["elapsed"], "unit": "rows/s", "ms_per_step": ores["elapsed"] / args.steps * 1e3,
                 "target_slots": orows, "per_rank": ores["per_rank"]}
        ores["shard"] = None
        shard = None

    if rank != 0:
        dist.destroy_process_group()
        return

    value = total_rows * args.steps / elapsed
    achieved = local_bytes / kern_avg_s / 1e9
    traffic, traffic_source = None, None
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if tfile.exists():       # PMC passes cannot run inside this process: taken from the committed profile of THIS build only
        try:
            ent = json.loads(tfile.read_text()).get(f"{args.workload}:{n_rows}x{n_cols}x{nnz_row}:k{k}", {})
            if ent.get("lib_source_sha") == lib_source_sha() and world == 1:
                traffic = ent.get("bytes_per_launch")
                traffic_source = (f"profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build (kernel sources sha "
                                  f"{ent.get('lib_source_sha')}), not of this process")
        except Exception:
            traffic = None
    par = f"row-sharded x{world} (distributed.partition_targets, contiguous cost-balanced slices)" + (", ONE gather of the packed slabs to rank 0 in the step" if world > 1 else "")
    out = {
        "metric": "similarity rows/sec, cosine k=100 on CSR" if args.workload == "c2" else f"similarity rows/sec, {sim_name} k={k} on CSR",
        "value": value,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload_txt,
            "target_slots": total_rows, "rows_per_gpu": total_rows // world, "cols": n_cols, "nnz_per_row": nnz_row, "k": k,
            "macs_per_row": macs / total_rows,
            "parallelism": par,
            "kept_entries_rank0": n_kept, "generic_windows_per_row": info["passes_total"] / max(1, main_res["per_rank"][0]["rows"]),
            "phase_share": phase_share(info),
            "per_rank": main_res["per_rank"],
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
            "kernel": dominant, "kernel_ms_avg": kern_avg_s * 1e3, "algorithmic_bytes_per_launch": local_bytes,
            "step_ms_avg_on_stream": float(np.mean(step_ms)), "sparse_kernel_ms": sparse_ms, "generic_kernel_ms": generic_ms,
        },
    }
    if other is not None:
        out["other_scaling"] = other
    if "_p3alpha_call" in extra_cfg:
        pres = measure(extra_cfg["_p3alpha_call"](), max(1, args.steps), 1, False)
        out["config"]["p3alpha_ms_per_step"] = pres["elapsed"] / max(1, args.steps) * 1e3
        pres["shard"] = None
    if world == 1 and not args.no_end_to_end:
        import similaripy_amd as sim
        sim.cosine(sp.csr_array(sp.random_array((2000, 500), density=0.02, format="csr", dtype=np.float32, random_state=np.random.default_rng(0))), k=10, verbose=False)   # (library / allocator warm-up)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            res = public_call[1](sim)
            ts.append(time.perf_counter() - t0)
        out["end_to_end_s"] = min(ts)
        out["end_to_end"] = {"call": public_call[0], "seconds": ts, "rows_per_s": n_rows / min(ts), "out_nnz": int(res.nnz)}
        del res
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


def physical_cores() -> int:
    """Distinct (package, core) pairs of the host; falls back to os.cpu_count()."""
    seen = set()
    try:
        for d in Path("/sys/devices/system/cpu").glob("cpu[0-9]*"):
            t = d / "topology"
            seen.add(((t / "physical_package_id").read_text().strip(), (t / "core_id").read_text().strip()))
    except Exception:
        seen = set()
    return len(seen) or (os.cpu_count() or 1)


_CPU_CHILD = r"""
import json, sys, time, types
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import splus_oracle as so
d = np.load(sys.argv[2], mmap_mode=None)
meta = json.loads(str(d["meta"]))
call = types.SimpleNamespace(**{k: d[k] for k in d.files if k != "meta"}, **meta["scalars"])
kind = "reference" if so.available("reference") else "port"
budget = float(sys.argv[3])
n_total = call.targets.shape[0]
def run(n, bs):
    c = types.SimpleNamespace(**vars(call)); c.targets = np.ascontiguousarray(call.targets[:n])
    t0 = time.perf_counter(); so.run_kernel(c, kind, num_threads=0, block_size=bs); return time.perf_counter() - t0
out = {"kind": kind, "threads": so.max_threads(kind)}
for name, bs in (("blocked_262144", 262144), ("unblocked", 0)):
    probe = min(n_total, 4000)
    run(min(n_total, 500), bs)                                  # thread team up, pages in
    rate = probe / run(probe, bs)
    sample = int(max(probe, min(n_total, rate * budget)))
    run(sample, bs)                                              # warm-up round at full sample size
    ts = [run(sample, bs) for _ in range(3)]
    r = [sample / t for t in ts]
    out[name] = {"rows_per_s": float(np.mean(r)), "std": float(np.std(r)), "rounds": r, "sample_rows": sample, "seconds": ts}
print(json.dumps(out))
"""


def cpu_baseline(call, round_s: float) -> dict:
    """The CPU kernel (oracle/_ref = the reference header compiled in place, else the C port) on a bounded prefix of the
    same target rows, in a child process whose OpenMP runtime is pinned to the physical cores."""
    import subprocess
    import tempfile

    cores = physical_cores()
    names = ("targets", "m1_data", "m1_indices", "m1_indptr", "m2_data", "m2_indices", "m2_indptr",
             "Xtversky", "Ytversky", "Xcosine", "Ycosine", "Xdepop", "Ydepop",
             "filter_m_indptr", "filter_m_indices", "target_col_m_indptr", "target_col_m_indices")
    scal = {n: getattr(call, n) for n in ("a1", "l1", "l2", "l3", "t1", "t2", "stabilized_shrink", "bayesian_shrink", "threshold",
                                          "k", "n_output_cols", "filter_mode", "target_col_mode")}
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        f = os.path.join(td, "call.npz")
        np.savez(f, meta=json.dumps({"scalars": scal}), **{n: getattr(call, n) for n in names})
        env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="spread", OMP_PLACES="cores")
        proc = subprocess.run([sys.executable, "-c", _CPU_CHILD, str(ROOT), f, str(round_s)], env=env, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("cpu_baseline child failed:\n" + proc.stderr[-2000:])
    r = json.loads(proc.stdout.strip().splitlines()[-1])
    b, u = r["blocked_262144"], r["unblocked"]
    log(f"cpu_baseline[{r['kind']}] {r['threads']} threads on {cores} physical cores: blocked(262144, the reference default) "
        f"{b['rows_per_s']:.0f} +- {b['std']:.0f} rows/s; unblocked {u['rows_per_s']:.0f} +- {u['std']:.0f} rows/s")
    return {
        "value": b["rows_per_s"], "std": b["std"], "unit": "rows/s", "cores": r["threads"], "kind": r["kind"],
        "sample": f"first {b['sample_rows']} of {call.targets.shape[0]} target rows of the same workload, the reference's default column "
                  f"blocking (block_size=0 -> 262144, s_plus.pyx:218-225; without its popularity reorder, which only permutes slot order), "
                  f"1 warm-up + 3 rounds of {np.mean(b['seconds']):.1f}s, {r['threads']} OpenMP threads = physical cores, "
                  f"OMP_PROC_BIND=spread OMP_PLACES=cores, dynamic schedule",
        "blocked_262144": b, "unblocked": u,
    }


if __name__ == "__main__":
    main()
