"""CPU tier: the row-sharded multi-process path over gloo (world_size 2), with the oracle as the compute
stand-in.  Checks the work-balanced partition and that the gathered result equals the single-process one."""
from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from similaripy_amd import _host                      # noqa: E402
from similaripy_amd import distributed as D           # noqa: E402


def _problem():
    rng = np.random.default_rng(11)
    m = sp.random_array((400, 120), density=0.08, format="csr", dtype=np.float32, random_state=rng).tolil()
    m[:40, :] = 0                                      # skew: the first rows are empty, the last are heavy
    m = sp.csr_array(m.tocsr())
    heavy = sp.random_array((30, 120), density=0.6, format="csr", dtype=np.float32, random_state=rng)
    m = sp.vstack([m, heavy]).tocsr()
    return _host.prepare(m, k=12, l2=1, target_rows=np.arange(0, 430, 1))


def test_partition_is_contiguous_and_work_balanced():
    call = _problem()
    w = D.row_work(call)
    for world in (1, 2, 3, 8):
        b = D.partition_targets(w, world)
        assert b[0] == 0 and b[-1] == call.n_targets and np.all(np.diff(b) >= 0)
        if world > 1:
            shares = np.array([w[b[r]:b[r + 1]].sum() for r in range(world)], dtype=np.float64)
            assert shares.max() <= 1.35 * shares.mean() + w.max()
    # by row count the split would be 215/215; by work the heavy tail pulls the boundary right
    assert D.partition_targets(w, 2)[1] > 215
    assert D.partition_targets(np.zeros(0), 4).tolist() == [0, 0, 0, 0, 0]


def test_both_routes_cut_the_same_bounds():
    """ONE partition cost model.  `distributed.row_cost` IS the library's (`sp_knn_target_costs`), so the slices the
    one-process-per-GPU route cuts (`partition_targets(row_cost(call), N)`) are the slices the in-library route (`n_devices = N`,
    `sp_knn_partition`) cuts — on a ratings-shaped item-item call (the configs[3] stand-in at a tenth of its size: skewed rows, heavy
    rows that the launch cuts into pieces and the model prices by their entries), host-built and device-built m2 alike."""
    from similaripy_amd import _abi
    from similaripy_amd.workloads import movielens_like_urm
    urm = movielens_like_urm(30_000, 40_000, 3_000_000, seed=3)
    m1 = sp.csr_array(urm.T.tocsr())
    for kw in (dict(), dict(m2_on_device=True)):
        call = _host.prepare(m1, k=50, **kw)
        cost = D.row_cost(call)
        macs = D.row_work(call).astype(np.float64) - 1.0
        assert cost.shape == (call.n_targets,) and np.all(cost >= macs)
        nnz1 = np.diff(call.m1_indptr)[call.targets]
        # generic rows pay 3 per output column; the heaviest rows also pay per entry (the launch cuts them into pieces)
        top = np.argsort(macs)[-5:]
        np.testing.assert_allclose(cost[top] - macs[top] - 3.0 * call.n_output_cols, 2100.0 * nnz1[top], rtol=1e-9)
        light = np.argsort(macs)[:5]
        assert np.all(cost[light] - macs[light] <= 30_000.0 + 1e-6)
        for world in (2, 4, 8):
            np.testing.assert_array_equal(D.partition_targets(cost, world), _abi.partition(call, world))
    # the prices are the library's: an override reaches both routes at once
    os.environ["SIMILARIPY_AMD_HEAVY_ENTRY_MACS"] = "0"
    try:
        c0 = D.row_cost(call)
        assert np.all(c0[top] - macs[top] == 3.0 * call.n_output_cols)
        np.testing.assert_array_equal(D.partition_targets(c0, 8), _abi.partition(call, 8))
    finally:
        del os.environ["SIMILARIPY_AMD_HEAVY_ENTRY_MACS"]
    with pytest.raises(_abi.HipLibraryError):
        bad = _host.prepare(m1, k=5, target_rows=[1, 2])
        bad.targets = np.array([1, 10 ** 6], dtype=np.int32)
        D.row_cost(bad)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import splus_oracle as so
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    call = _problem()

    def compute(c):
        rows, cols, vals = so.run_kernel(c, "port", num_threads=1)
        counts = so.slot_counts(rows, cols, vals, c.targets, c.k)[0] if c.n_targets else np.zeros(0, np.int32)
        return rows, cols, vals, counts

    out = D.sharded_knn(call, compute, dst=0)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_single_process_gloo():
    from oracle import splus_oracle as so
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows, cols, vals, counts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    call = _problem()
    want = so.run_kernel(call, "port", num_threads=1)
    k = call.k
    got_c = so.canonical(rows, cols, vals, call.targets, k)
    want_c = so.canonical(*want, call.targets, k)
    for (gc, gv), (wc, wv) in zip(got_c, want_c):
        np.testing.assert_array_equal(gc, wc)
        np.testing.assert_array_equal(gv, wv)
    np.testing.assert_array_equal(counts, so.slot_counts(*want, call.targets, k)[0])
    # rows array follows the slot convention
    r = rows.reshape(-1, k)
    assert np.all((r == call.targets[:, None]) | (r == 0))


def test_compact_slice_is_the_same_problem():
    """slice_call(compact=True) — what a rank uploads — gives the oracle the same answers as the plain slice, MATRIX
    selectors and X* vectors included."""
    from oracle import splus_oracle as so
    rng = np.random.default_rng(3)
    m = sp.random_array((300, 90), density=0.1, format="csr", dtype=np.float32, random_state=rng)
    m2 = sp.random_array((90, 150), density=0.1, format="csr", dtype=np.float32, random_state=rng)
    filt = sp.random_array((300, 150), density=0.05, format="csr", dtype=np.float32, random_state=rng)
    call = _host.prepare(m, m2, k=9, l1=0.4, l2=0.6, stabilized_shrink=2.0, filter_cols=filt,
                         target_rows=np.arange(20, 290, dtype=np.int32))
    assert call.filter_mode == _host.MODE_MATRIX
    for lo, hi in ((0, 100), (100, 270), (37, 38)):
        a = D.slice_call(call, lo, hi)
        b = D.slice_call(call, lo, hi, compact=True)
        assert b.n_rows_m1 == hi - lo and b.m1_indptr[0] == 0 and b.targets.min() == 0
        ra, ca, va = so.run_kernel(a, "port", num_threads=1)
        rb, cb, vb = so.run_kernel(b, "port", num_threads=1)
        np.testing.assert_array_equal(ca, cb)
        np.testing.assert_array_equal(va, vb)
        # rows of the compact call are shifted by the slice's first row
        real = (ra != 0) | (ca != 0) | (va != 0)
        np.testing.assert_array_equal(ra[real], rb[real] + int(a.targets.min()))


def test_row_work_of_a_device_transposed_call():
    """A KernelCall of the public wrappers (m2 = m1^T left to the device) has no m2 arrays: the work vector comes from
    the column counts of m1 and equals the one of the host-transposed call."""
    rng = np.random.default_rng(5)
    m = sp.random_array((200, 70), density=0.1, format="csr", dtype=np.float32, random_state=rng)
    host = _host.prepare(m, k=5)
    dev = _host.prepare(m, k=5, m2_on_device=True)
    assert dev.m2_is_m1t and dev.m2_indptr.size == 0
    np.testing.assert_array_equal(D.row_work(host), D.row_work(dev))



@pytest.mark.gpu
def test_sharded_device_problem_nccl_world1_equals_single_call():
    """The shipped multi-GPU driver (ShardedDeviceProblem: partition -> resident slice -> kernel -> gather) under the
    nccl (= RCCL) backend with one rank equals the plain single call."""
    import torch
    import torch.distributed as dist
    from oracle import splus_oracle as so
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(7)
        m = sp.random_array((3000, 2500), density=0.01, format="csr", dtype=np.float32, random_state=rng)
        call = _host.prepare(m, k=20, l2=1, target_rows=np.arange(100, 2900, dtype=np.int32))
        sh = D.ShardedDeviceProblem(call)
        assert sh.world == 1 and sh.n_loc == call.n_targets
        sh.run()
        rows, cols, vals, counts = sh.result()
        r1, c1, v1, n1 = _host.run_hip(call)
        k = call.k
        so.compare_topk(so.canonical(rows, cols, vals, call.targets, k), so.canonical(r1, c1, v1, call.targets, k), k, rtol=1e-6, atol=0, what="sharded vs single")
        np.testing.assert_array_equal(counts, n1)
        want = so.canonical(*so.run_kernel(call, "port"), call.targets, k)
        so.compare_topk(so.canonical(rows, cols, vals, call.targets, k), want, k, rtol=1e-5, atol=1e-7, what="sharded vs oracle")
        # the RCCL calls of the N > 1 step on this one-GPU box: a group of one rank that still gathers — three sub-launches, sub-slab j
        # sent with an asynchronous gather on the communication stream behind sub-launch j, the step closed by the stream-level waits
        sh3 = D.ShardedDeviceProblem(call, phases=3, gather_alone=True)
        assert sh3.phases == 3 and sh3.comm_stream is not None and sh3.recv is not None and not sh3.host_gather
        for _ in range(3):
            sh3.run()
        r3, c3, v3, n3 = sh3.result()
        so.compare_topk(so.canonical(r3, c3, v3, call.targets, k), want, k, rtol=1e-5, atol=1e-7, what="split-phase gather over nccl vs oracle")
        np.testing.assert_array_equal(n3, n1)
        # hip_compute() defaults to the rank's own device, and sharded_knn puts its gather tensors there
        out = D.sharded_knn(call, D.hip_compute())
        np.testing.assert_array_equal(out[3], n1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("runner", ["oracle_runner", "oracle_runner_root_free"])
@pytest.mark.parametrize("fmt,chunk", [("csr", None), ("csr", 100), ("coo", 64)])
def test_multi_gpu_entry_spawns_ranks_and_assembles_gloo(fmt, chunk, runner):
    """similaripy_amd.multi_gpu.run_call — the one-process entry that spawns a worker per device — over gloo with the
    oracle kernel (tests/mgpu_oracle_runner.py): shm hand-over, chunked streaming, CSR / COO assembly equal the
    single-process result.  Both delivery protocols: the chunk gathered on rank 0, and ROOT-FREE — every rank hands the parent its
    own slots (what multi_gpu._hip_runner does by default since round 6: each rank's slab comes down over its own PCIe link)."""
    from oracle import splus_oracle as so
    from similaripy_amd import multi_gpu
    call = _problem()
    res = multi_gpu.run_call(call, devices=2, format_output=fmt, chunk_rows=chunk, backend="gloo", runner=f"tests.mgpu_oracle_runner:{runner}")
    rows, cols, vals = so.run_kernel(call, "port", num_threads=1)
    counts = so.slot_counts(rows, cols, vals, call.targets, call.k)[0]
    want = _host.finish(call, rows, cols, vals, counts, fmt)
    assert type(res) is type(want) and res.shape == want.shape and res.nnz == want.nnz
    a, b = res.tocsr(), want.tocsr()
    a.sort_indices()
    b.sort_indices()
    np.testing.assert_array_equal(a.indptr, b.indptr)
    np.testing.assert_array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)


def _rows_of(res):
    res = res.tocsr()
    out = []
    for t in range(res.shape[0]):
        c, v = res.indices[res.indptr[t]:res.indptr[t + 1]], res.data[res.indptr[t]:res.indptr[t + 1]]
        o = np.argsort(c, kind="stable")
        out.append((c[o].astype(np.int32), v[o].astype(np.float32)))
    return out


def _same_topk(got, want, k, what):
    """Two CSR results row by row with the tie-aware comparator of the parity tests (a k-th place tie may resolve differently)."""
    from oracle import splus_oracle as so
    so.compare_topk(_rows_of(got), _rows_of(want), k, rtol=1e-5, atol=1e-7, what=what)


def _csr_of(triples, call):
    from oracle import splus_oracle as so
    rows, cols, vals = triples
    counts = so.slot_counts(rows, cols, vals, call.targets, call.k)[0]
    return _host.finish(call, rows, cols, vals, counts, "csr")


@pytest.mark.gpu
def test_multi_gpu_entry_one_device_nccl():
    """The spawned route on the real thing: one worker on device 0 under nccl (ShardedDeviceProblem, resident operands,
    chunked), through the public wrappers — explicit `multi_gpu.similarity(...)` and the SIMILARIPY_AMD_DEVICES route."""
    import similaripy_amd as sim
    from oracle import splus_oracle as so
    rng = np.random.default_rng(13)
    m = sp.random_array((6000, 900), density=0.01, format="csr", dtype=np.float32, random_state=rng)
    want = sim.cosine(m, k=15, verbose=False, format_output="csr")
    for chunk in (None, 2500):
        got = sim.multi_gpu.similarity("cosine", m, k=15, verbose=False, format_output="csr", devices=[0], chunk_rows=chunk)
        assert got.shape == want.shape and got.nnz == want.nnz
        _same_topk(got, want, 15, "spawned route vs single process")
    # ... and against the oracle, row by row (tie-aware)
    call = _host.prepare(m, k=15, l2=1)
    _same_topk(got, _csr_of(so.run_kernel(call, "port"), call), 15, "spawned route vs oracle")
    # rp3beta: preprocessing on the host side of the route
    r1 = sim.multi_gpu.similarity("rp3beta", m, alpha=0.8, beta=0.4, k=10, verbose=False, format_output="csr", devices=[0])
    r0 = sim.rp3beta(m, alpha=0.8, beta=0.4, k=10, verbose=False, format_output="csr")
    assert r1.nnz == r0.nnz
    _same_topk(r1, r0, 10, "rp3beta: spawned route vs single process")


# ---- the HIP kernels at world_size 2 on the ONE GPU of the test box ---------------------------------------------------------------
def _hip_problem(which: str):
    """Problems for the two-rank HIP test: both row kernels, MATRIX selectors, a target_rows subset, skewed work."""
    rng = np.random.default_rng(21)
    if which == "sparse+filter":          # wide output, light rows: the sparse row kernel (monotone variant) with a MATRIX filter
        m = sp.random_array((30000, 1500), density=0.004, format="csr", dtype=np.float32, random_state=rng)
        filt = sp.random_array((30000, 30000), density=20.0 / 30000, format="csr", dtype=np.float32, random_state=rng)
        tg = np.sort(rng.choice(30000, size=9000, replace=False)).astype(np.int32)
        return _host.prepare(m, k=20, l2=1.0, c1=0.5, c2=0.5, filter_cols=filt, target_rows=tg)
    if which == "general+target":         # Tversky + shrink (general variant, judge) with a MATRIX target selector, explicit m2
        m = sp.random_array((12000, 900), density=0.01, format="csr", dtype=np.float32, random_state=rng)
        m2 = sp.random_array((900, 20000), density=0.01, format="csr", dtype=np.float32, random_state=rng)
        tcols = sp.random_array((12000, 20000), density=300.0 / 20000, format="csr", dtype=np.float32, random_state=rng)
        return _host.prepare(m, m2, k=15, l1=0.6, l2=0.4, t1=0.7, t2=0.3, stabilized_shrink=2.0, target_cols=tcols,
                             target_rows=np.arange(500, 11500, dtype=np.int32))
    # dense rows over few columns: the generic row kernel; the last rows are much heavier (the partition is by work)
    # (the heavy rows hold ones only: sums of 1500 products are then exact in any order — the oracle adds sequentially, the device
    # does not, and 1e-5 is not a bar a 1500-term float32 sum of random values can be held to)
    m = sp.random_array((2500, 3000), density=0.02, format="csr", dtype=np.float32, random_state=rng).tolil()
    m[2300:, :] = 0.0
    m[2300:, :1500] = 1.0
    return _host.prepare(sp.csr_array(m.tocsr()), k=30, l2=1.0)


def _hip_worker(rank, world, port, q, which):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SIMILARIPY_AMD_DEVICE"] = "0"
    import datetime
    import traceback
    torch.cuda.set_device(0)                                   # both ranks on the one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    try:
        call = _hip_problem(which)
        out = {}
        # (1) host-array driver with the HIP library as compute
        out["sharded_knn"] = D.sharded_knn(call, D.hip_compute(device=0), dst=0)
        # (2) the shipped class: compact slice resident on the device, slabs through the host for the gloo gather;
        #     one launch + one gather, and the split-phase form (3 sub-launches, sub-slab j gathered behind sub-launch j)
        for phases in (1, 3):
            sh = D.ShardedDeviceProblem(call, device=torch.device("cuda", 0), phases=phases, persist_prep=(phases == 3))      # (the second step then reuses the first one's passes over m2)
            assert sh.world == 2 and sh.host_gather and sh.phases == phases
            assert sh.prob.call.n_rows_m1 <= call.n_rows_m1              # compact: only the slice's own rows of m1 went up
            sh.run()
            sh.run()                                                      # (a second step reuses the resident problem and its workspace)
            out[f"sdp{phases}"] = sh.result()
            out[f"n_loc{phases}"] = sh.n_loc
        if rank == 0:
            q.put(out)
        else:
            assert all(v is None for k_, v in out.items() if not k_.startswith("n_loc"))
            q.put({"n_loc": out["n_loc1"]})
        dist.barrier()
    except Exception:                                          # (reported to the parent: a dead rank must not leave the other one waiting)
        q.put({"error": f"rank {rank}: {traceback.format_exc()[-1500:]}"})
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["sparse+filter", "general+target", "generic+skew"])
def test_hip_kernels_at_world_size_2_on_one_gpu(which):
    """the HIP path had never run at world_size >= 2.  Two spawned ranks over gloo, both on device 0: `sharded_knn` with
    `hip_compute`, and `ShardedDeviceProblem` (compact slice resident, per-rank `partition_targets` slices on the HIP kernels, slabs
    padded to n_max, split-phase sub-slabs) against the ORACLE on the whole problem (s_plus.h:313, 337: the row loop that is sharded)."""
    from oracle import splus_oracle as so
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_worker, args=(r, world, port, q, which)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(world):
            got.append(q.get(timeout=200))
            assert "error" not in got[-1], got[-1]["error"]
    finally:
        for p in procs:
            p.join(timeout=30 if len(got) == world and "error" not in got[-1] else 1)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    root = next(g for g in got if "sharded_knn" in g)
    other = next(g for g in got if "sharded_knn" not in g)
    call = _hip_problem(which)
    k = call.k
    assert 0 < root["n_loc1"] < call.n_targets and root["n_loc1"] + other["n_loc"] == call.n_targets       # both ranks had rows
    want_raw = so.run_kernel(call, "port")
    want = so.canonical(*want_raw, call.targets, k)
    wcnt = so.slot_counts(*want_raw, call.targets, k)[0]
    for key in ("sharded_knn", "sdp1", "sdp3"):
        rows, cols, vals, counts = root[key]
        so.compare_topk(so.canonical(rows, cols, vals, call.targets, k), want, k, rtol=1e-5, atol=1e-7, what=f"{which}: {key} at world 2 vs oracle")
        np.testing.assert_array_equal(counts, wcnt)
        r = rows.reshape(-1, k)
        assert np.all((r == call.targets[:, None]) | (r == 0))


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_over_gloo_on_one_gpu():
    """`python bench.py --gpus 2` started plain launches its own ranks (torch.distributed.run on 127.0.0.1): the path the driver's N > 1
    runs take, here with two ranks sharing the box's one GPU over gloo (functional check)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--rows", "60000", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-end-to-end", "--no-traffic", "--no-other-workloads"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run(cmd, cwd=str(root), capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and out["steps"] == 2
    per = out["config"]["per_rank"]
    assert len(per) == 2 and all(p["rows"] > 0 for p in per) and sum(p["rows"] for p in per) == 60000
    assert out.get("gather_exposed_ms") is not None and out["value"] > 0
    # one step's results to HOST memory, both ways (round 6): over every rank's own link, and through the root
    r2h = out["result_to_host_ms"]
    assert r2h["own_link"] > 0 and r2h["via_root"] > 0
    # every world size times the same work: the per-call passes over m2 are redone by every step
    assert out["config"]["persist_prep"] is False and out["config"]["m2_prep"] == "every step"
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["value"] > 0
