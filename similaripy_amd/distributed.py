"""Row-sharded multi-GPU execution: one process per GPU, torch.distributed as plumbing.

The path shards exactly like the reference's only parallel loop — output rows are independent
(`#pragma omp for` over targets, s_plus.h:337; the public `target_rows` kwarg).  Layout:

  * `targets` is cut into `world_size` CONTIGUOUS slices balanced by work
    (MACs(t) = sum_u nnz(m2 row u), the quantity the kernel itself schedules on) — not by row count,
    so skewed matrices (power-law item popularity) stay balanced;
  * m2 and the Y* vectors are replicated on every GPU, m1 / X* / selectors are indexed by absolute
    row id so every rank can hold them whole (they are small next to m2);
  * no collective during compute; ONE gather of the (cols, values, counts) slabs at the end
    (RCCL over xGMI when the backend is "nccl", gloo on CPU for tests).  Slabs are padded to the
    largest slice so a plain `gather` works; `rows` is implied by the slot and rebuilt on the root.

`compute` is injected (a callable `(call_slice) -> rows, cols, values, counts`) so the CPU test tier can
drive this module over gloo with the oracle kernel; the product passes the HIP path.
"""
from __future__ import annotations

import copy
from typing import Callable, Optional, Tuple

import numpy as np

from ._host import KernelCall


def row_work(call: KernelCall) -> np.ndarray:
    """MACs per target slot: sum over the row's m1 entries of the length of the m2 row they select."""
    nnz2 = np.diff(call.m2_indptr).astype(np.int64)
    per_entry = nnz2[call.m1_indices]
    csum = np.concatenate(([0], np.cumsum(per_entry)))
    macs_row = csum[call.m1_indptr[1:]] - csum[call.m1_indptr[:-1]]
    return macs_row[call.targets] + 1          # +1: an empty row still costs a slot


def partition_targets(work: np.ndarray, world_size: int) -> np.ndarray:
    """Boundaries b[0..world_size] of contiguous slices of equal cumulative work (b[0]=0, b[-1]=n)."""
    n = int(work.shape[0])
    csum = np.cumsum(work, dtype=np.float64)
    total = float(csum[-1]) if n else 0.0
    bounds = np.zeros(world_size + 1, dtype=np.int64)
    for r in range(1, world_size):
        bounds[r] = int(np.searchsorted(csum, total * r / world_size, side="left")) if n else 0
    bounds[world_size] = n
    return np.maximum.accumulate(bounds)


def slice_call(call: KernelCall, lo: int, hi: int) -> KernelCall:
    """The same problem restricted to target slots [lo, hi)."""
    c = copy.copy(call)
    c.targets = np.ascontiguousarray(call.targets[lo:hi])
    return c


def sharded_knn(call: KernelCall, compute: Callable[[KernelCall], Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]],
                dst: int = 0, group=None, device=None):
    """Run `compute` on this rank's slice of `call.targets` and gather everything on `dst`.

    Every rank passes the same `call` (same targets, replicated operands).  Returns
    (rows, cols, values, counts) for ALL targets on rank `dst`, None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    k = call.k
    bounds = partition_targets(row_work(call), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    n_max = int(np.max(np.diff(bounds))) if world else 0

    rows, cols, vals, counts = compute(slice_call(call, lo, hi))
    n_loc = hi - lo

    dev = torch.device("cpu") if device is None else torch.device(device)
    pad_cols = torch.zeros(n_max * k, dtype=torch.int32, device=dev)
    pad_vals = torch.zeros(n_max * k, dtype=torch.float32, device=dev)
    pad_cnt = torch.zeros(n_max, dtype=torch.int32, device=dev)
    pad_cols[: n_loc * k] = torch.as_tensor(np.ascontiguousarray(cols), device=dev)
    pad_vals[: n_loc * k] = torch.as_tensor(np.ascontiguousarray(vals), device=dev)
    pad_cnt[:n_loc] = torch.as_tensor(np.ascontiguousarray(counts), device=dev)

    gather_c = [torch.empty_like(pad_cols) for _ in range(world)] if rank == dst else None
    gather_v = [torch.empty_like(pad_vals) for _ in range(world)] if rank == dst else None
    gather_n = [torch.empty_like(pad_cnt) for _ in range(world)] if rank == dst else None
    dist.gather(pad_cols, gather_c, dst=dst, group=group)
    dist.gather(pad_vals, gather_v, dst=dst, group=group)
    dist.gather(pad_cnt, gather_n, dst=dst, group=group)
    if rank != dst:
        return None

    n = call.n_targets
    out_cols = np.zeros(n * k, dtype=np.int32)
    out_vals = np.zeros(n * k, dtype=np.float32)
    out_cnt = np.zeros(n, dtype=np.int32)
    for r in range(world):
        a, b = int(bounds[r]), int(bounds[r + 1])
        out_cols[a * k: b * k] = gather_c[r][: (b - a) * k].cpu().numpy()
        out_vals[a * k: b * k] = gather_v[r][: (b - a) * k].cpu().numpy()
        out_cnt[a:b] = gather_n[r][: b - a].cpu().numpy()
    # rows: slot i holds targets[i] in its first counts[i] entries, 0 in the padding
    real = (np.arange(k, dtype=np.int32)[None, :] < out_cnt[:, None])
    out_rows = np.where(real, call.targets[:, None], 0).astype(np.int32).ravel()
    return out_rows, out_cols, out_vals, out_cnt


def hip_compute(device: Optional[int] = None, **tuning):
    """`compute` callable for sharded_knn backed by the HIP library (the product path)."""
    from . import _host

    def run(call_slice: KernelCall):
        return _host.run_hip(call_slice, device=device, **tuning)

    return run
