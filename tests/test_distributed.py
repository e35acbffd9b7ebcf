"""CPU tier: the row-sharded multi-process path over gloo (world_size 2), with the oracle as the compute
stand-in.  Checks the work-balanced partition and that the gathered result equals the single-process one."""
from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import numpy as np
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
