"""CPU-tier stand-in for multi_gpu._hip_runner: the same generator protocol (chunks of the target list, every chunk cut
`world` ways by work, one gather per chunk) over gloo with the oracle kernel as the compute.  Test infrastructure only."""
from __future__ import annotations

import numpy as np


def oracle_runner(call, group, device, chunk_rows):
    import torch.distributed as dist
    from oracle import splus_oracle as so
    from similaripy_amd import distributed as D

    def compute(c):
        rows, cols, vals = so.run_kernel(c, "port", num_threads=1)
        counts = so.slot_counts(rows, cols, vals, c.targets, c.k)[0] if c.n_targets else np.zeros(0, np.int32)
        return rows, cols, vals, counts

    n = call.n_targets
    chunk = n if not chunk_rows else int(chunk_rows)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        out = D.sharded_knn(D.slice_call(call, lo, hi), compute, dst=0, group=group)
        yield (lo, hi, out[1], out[2], out[3]) if dist.get_rank(group) == 0 else None


def oracle_runner_root_free(call, group, device, chunk_rows):
    """The root-free protocol of multi_gpu._hip_runner (round 6): every rank yields its OWN slots (first, one past the last, cols, values,
    counts) of every chunk — cut by the same cost-balanced partition — and nothing is gathered."""
    import torch.distributed as dist
    from oracle import splus_oracle as so
    from similaripy_amd import distributed as D

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = call.n_targets
    chunk = n if not chunk_rows else int(chunk_rows)
    cost = D.row_cost(call)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        b = D.partition_targets(cost[lo:hi], world) + lo
        a0, a1 = int(b[rank]), int(b[rank + 1])
        if a1 <= a0:
            yield None
            continue
        c = D.slice_call(call, a0, a1)
        rows, cols, vals = so.run_kernel(c, "port", num_threads=1)
        counts = so.slot_counts(rows, cols, vals, c.targets, c.k)[0]
        yield (a0, a1, cols, vals, counts)
