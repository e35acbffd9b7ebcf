"""Row-sharded multi-GPU entry for callers that are ONE Python process: spawns one worker process per GPU.

    import similaripy_amd as sim
    S = sim.multi_gpu.similarity("cosine", urm.T, k=100, devices=[0, 1, 2, 3, 4, 5, 6, 7], format_output="csr")
    R = sim.multi_gpu.similarity("dot_product", urm, W.T, k=100, filter_cols=urm, devices=8, chunk_rows=1_000_000)

or, without touching the call sites, `SIMILARIPY_AMD_DEVICES=0,1,2,3` in the environment makes every public wrapper
(`sim.cosine`, ...) take this route (see `_host._s_plus_impl`).

What runs (BASELINE configs[3]/[4]): the parent does the host stages of s_plus.pyx once (`_host.prepare`) and
parks the kernel's operands in /dev/shm (raw .npy, memory-mapped by the workers; rank 0 hands the results back through shared-memory
segments, one per array); every worker maps them, joins a `torch.distributed` group (backend "nccl" = RCCL
over xGMI), and runs `distributed.ShardedDeviceProblem` — `partition_targets` (contiguous, work-balanced), its slice of m1
plus the replicated m2 / Y* resident on its GPU, the kernel, ONE gather of the (cols, values, counts) slabs to rank 0 —
the same class `bench.py --gpus N` measures.  `chunk_rows` streams the target list in chunks (one gather per chunk), so
that 10M users x k=100 never need 12 bytes x 10^9 of host arrays at once: rank 0 turns every gathered chunk into its CSR
rows right away.  The parent assembles the scipy result exactly as the single-GPU path does.

Workers are spawned (`torch.multiprocessing`, start method "spawn"); starting them and RCCL costs seconds, so this pays for
jobs of 10^6 rows and more.  A user who already runs one process per GPU (torchrun) uses `ShardedDeviceProblem` directly.
"""
from __future__ import annotations

import importlib
import json
import os
import shutil
import socket
import tempfile
from typing import Optional, Sequence, Union

import numpy as np
import scipy.sparse as sp

from . import _host
from ._host import KernelCall

_ARRAYS = ("targets", "m1_data", "m1_indices", "m1_indptr", "m2_data", "m2_indices", "m2_indptr",
           "Xtversky", "Ytversky", "Xcosine", "Ycosine", "Xdepop", "Ydepop",
           "filter_m_indptr", "filter_m_indices", "target_col_m_indptr", "target_col_m_indices")
_SCALARS = ("n_rows_m1", "n_rows_m2", "n_output_cols", "k", "a1", "l1", "l2", "l3", "t1", "t2", "stabilized_shrink", "bayesian_shrink",
            "threshold", "filter_mode", "target_col_mode", "m2_is_m1t", "p3_alpha", "depop_rowsum_p2")


def devices_from_env() -> Optional[list]:
    v = os.environ.get("SIMILARIPY_AMD_DEVICES", "").strip()
    if not v:
        return None
    d = [int(x) for x in v.split(",") if x.strip() != ""]
    return d if len(d) > 1 else None


def _save_call(call: KernelCall, d: str) -> None:
    for n in _ARRAYS:
        np.save(os.path.join(d, n + ".npy"), np.ascontiguousarray(getattr(call, n)))
    with open(os.path.join(d, "scalars.json"), "w") as f:
        json.dump({n: getattr(call, n) for n in _SCALARS}, f)


def _load_call(d: str) -> KernelCall:
    arrays = {n: np.load(os.path.join(d, n + ".npy"), mmap_mode="r") for n in _ARRAYS}
    with open(os.path.join(d, "scalars.json")) as f:
        scal = json.load(f)
    return KernelCall(**{n: np.asarray(a) for n, a in arrays.items()}, **scal)


def _shm_put(arr: np.ndarray) -> dict:
    """Worker: one result array into a POSIX shared-memory segment of its own (ONE copy, no container format); returns what the
    parent needs to map it.  The segment outlives this process: the parent unlinks it (`_shm_take`)."""
    from multiprocessing import resource_tracker, shared_memory

    a = np.ascontiguousarray(arr)
    seg = shared_memory.SharedMemory(create=True, size=max(1, a.nbytes))
    np.ndarray(a.shape, dtype=a.dtype, buffer=seg.buf)[...] = a
    meta = {"name": seg.name, "dtype": a.dtype.str, "shape": list(a.shape)}
    try:      # (the segment is the parent's to remove: this process's tracker must not unlink it at exit)
        resource_tracker.unregister(seg._name, "shared_memory")
    except Exception:      # noqa: BLE001
        pass
    seg.close()
    return meta


def _shm_take(meta: dict) -> np.ndarray:
    """Parent: the array of a worker's segment (copied out of the mapping), the segment removed."""
    from multiprocessing import shared_memory

    seg = shared_memory.SharedMemory(name=meta["name"])
    try:
        return np.ndarray(tuple(meta["shape"]), dtype=np.dtype(meta["dtype"]), buffer=seg.buf).copy()
    finally:
        seg.close()
        try:
            seg.unlink()
        except FileNotFoundError:
            pass


def _shm_drop(meta: dict) -> None:
    from multiprocessing import shared_memory
    try:
        seg = shared_memory.SharedMemory(name=meta["name"])
        seg.close()
        seg.unlink()
    except Exception:      # noqa: BLE001
        pass


def _hip_runner(call: KernelCall, group, device: int, chunk_rows: Optional[int]):
    """Generator on every rank: yields (lo, hi, cols, values, counts) of target slots [lo, hi) on rank 0, None elsewhere."""
    import torch

    from .distributed import ShardedDeviceProblem

    torch.cuda.set_device(device)
    n = call.n_targets
    chunk = n if not chunk_rows else int(chunk_rows)
    # the operands go up once; every chunk is the same resident problem with another slice of the target list
    shard = ShardedDeviceProblem(call, group=group, device=torch.device("cuda", device), chunk_rows=chunk)
    root_free = os.environ.get("SIMILARIPY_AMD_GATHER_TO_ROOT", "0") in ("", "0")
    for lo, hi in shard.chunks():
        if root_free:
            # every rank brings its OWN slots to the host over its own PCIe link and hands them to the parent itself (round 6): no slab
            # crosses xGMI only to queue at the root's one link (the RCCL gather — SIMILARIPY_AMD_GATHER_TO_ROOT=1, and what
            # `bench.py --gpus N` times — is the form that leaves the results resident on one device)
            shard.run_chunk(lo, hi, gather=False)
            a0, a1, cols, vals, cnt = shard.own_result(lo, hi)
            yield (a0, a1, cols, vals, cnt) if a1 > a0 else None
        else:
            shard.run_chunk(lo, hi)
            out = shard.chunk_result(lo, hi)
            yield (lo, hi) + tuple(out) if out is not None else None


def _worker(rank: int, world: int, port: int, shm: str, backend: str, runner: str, devices: Sequence[int], chunk_rows: Optional[int], csr: bool):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(devices[rank])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {}
    if backend == "nccl":
        import torch
        torch.cuda.set_device(devices[rank])
        kw["device_id"] = torch.device("cuda", devices[rank])
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        call = _load_call(shm)
        mod, fn = runner.split(":")
        run = getattr(importlib.import_module(mod), fn)
        pieces = []
        for item in run(call, None, devices[rank], chunk_rows):
            if item is None:      # (a runner that gathers yields the chunk on rank 0 only; a root-free one yields every rank's own slots)
                continue
            lo, hi, cols, vals, counts = item
            if csr:
                # every gathered chunk becomes its CSR rows at once (padding and zeros dropped): 8 bytes per kept entry
                piece = _host.build_csr(call.targets[lo:hi], cols, vals, counts, call.k, call.n_rows_m1, call.n_output_cols)
                pieces.append((lo, hi, piece.indptr.astype(np.int64), piece.indices, piece.data))
            else:
                pieces.append((lo, hi, cols, vals, counts))
        # results go back through shared-memory segments (one per array: mapped by the parent, no .npz container to write and parse);
        # every rank hands over its own pieces
        with open(os.path.join(shm, f"out_{rank}.json"), "w") as f:
            json.dump({"pieces": [{"lo": int(p[0]), "hi": int(p[1]), "a": _shm_put(p[2]), "b": _shm_put(p[3]), "c": _shm_put(p[4])} for p in pieces]}, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_call(call: KernelCall, devices: Union[int, Sequence[int]], format_output: str = "csr", chunk_rows: Optional[int] = None,
             backend: str = "nccl", runner: str = "similaripy_amd.multi_gpu:_hip_runner"):
    """The kernel stage of a prepared call on several GPUs; returns the scipy result (what `_host.finish` returns)."""
    import torch.multiprocessing as mp

    devs = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
    if not devs:
        raise ValueError("devices is empty")
    t = call.targets
    increasing = call.n_targets <= 1 or bool(np.all(t[1:] > t[:-1]))
    csr = format_output == "csr" and increasing       # (repeated / unsorted target rows: the slots are assembled in one piece)
    shm = tempfile.mkdtemp(prefix="similaripy_amd_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        _save_call(call, shm)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_worker, args=(len(devs), port, shm, backend, runner, devs, chunk_rows, csr), nprocs=len(devs), join=True)
        metas = []
        for r in range(len(devs)):
            with open(os.path.join(shm, f"out_{r}.json")) as f:
                metas += json.load(f)["pieces"]
        pieces = []
        try:
            for m in metas:
                pieces.append((int(m["lo"]), int(m["hi"]), _shm_take(m["a"]), _shm_take(m["b"]), _shm_take(m["c"])))
        except BaseException:
            for m in metas:
                for key in ("a", "b", "c"):
                    _shm_drop(m[key])
            raise
    finally:
        shutil.rmtree(shm, ignore_errors=True)
    pieces.sort(key=lambda p: p[0])
    n, k = call.n_targets, call.k
    if csr:
        # the chunks' row pointer arrays add up (a chunk holds entries only in the rows of its own targets)
        indptr = np.zeros(call.n_rows_m1 + 1, dtype=np.int64)
        for _, _, ip, _, _ in pieces:
            indptr += ip
        # entries: chunk after chunk = row order, because the targets ascend
        indices = np.concatenate([p[3] for p in pieces]) if pieces else np.zeros(0, np.int32)
        data = np.concatenate([p[4] for p in pieces]) if pieces else np.zeros(0, np.float32)
        idx_dtype = np.int32 if max(int(indptr[-1]), call.n_output_cols) <= np.iinfo(np.int32).max else np.int64
        return sp.csr_array((data, indices.astype(idx_dtype, copy=False), indptr.astype(idx_dtype)), shape=(call.n_rows_m1, call.n_output_cols), dtype=np.float32)
    cols = np.zeros(n * k, dtype=np.int32)
    vals = np.zeros(n * k, dtype=np.float32)
    counts = np.zeros(n, dtype=np.int32)
    for lo, hi, c, v, cnt in pieces:
        cols[lo * k: hi * k], vals[lo * k: hi * k], counts[lo:hi] = c, v, cnt
    real = np.arange(k, dtype=np.int32)[None, :] < counts[:, None]
    rows = np.where(real, call.targets[:, None], 0).astype(np.int32).ravel()
    return _host.finish(call, rows, cols, vals, counts, format_output)


def similarity(name: str, matrix1, matrix2=None, *, devices: Union[int, Sequence[int]], chunk_rows: Optional[int] = None,
               mode: Optional[str] = None, **kwargs):
    """`similaripy_amd.<name>(matrix1, matrix2, **kwargs)` with the kernel stage sharded over `devices`.

    mode "threads" (default without chunk_rows): the library's multi-device call (sp_knn_args.n_devices / device_ids, ABI 5) —
    this process, one host thread per device, no spawn, no process group, every device-side stage of the single-GPU path.
    mode "processes" (default with chunk_rows): one spawned worker per GPU, RCCL gather (run_call)."""
    from . import similarity as S

    fn = getattr(S, name)
    mode = mode or ("processes" if chunk_rows else "threads")
    if mode not in ("threads", "processes"):
        raise ValueError("mode must be 'threads' or 'processes'")
    token = _Route(devices, chunk_rows, mode)
    _host._MULTI_GPU_ROUTE.append(token)
    try:
        return fn(matrix1, matrix2, **kwargs)
    finally:
        _host._MULTI_GPU_ROUTE.remove(token)


class _Route:
    def __init__(self, devices, chunk_rows, mode="threads"):
        self.devices, self.chunk_rows, self.mode = devices, chunk_rows, mode
