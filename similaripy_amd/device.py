"""Device-resident form of the kernel call.

`_host.run_hip` is the drop-in path (host buffers in, host buffers out, copies inside the C
library).  Benchmarks, pipelines and the multi-GPU driver instead keep the CSR operands resident in
HBM and launch on a stream of their choice; torch is used here purely as plumbing — device
memory (tensors), streams, and later torch.distributed — the compute is the same C-ABI entry
point (`sp_knn_f32_i32` with on_device=1).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _abi
from ._host import KernelCall


def _torch():
    import torch
    return torch


class DeviceProblem:
    """All operands of one similarity problem, resident on one GPU.

    m2 and the Y* vectors are what gets replicated on every GPU in the multi-GPU layout
    (SURVEY §8e); m1 / X* / selectors are indexed by absolute row id, so a rank that only
    handles a slice of `targets` can still hold them whole (they are small next to m2) or hold
    a row-sliced copy made with `KernelCall` slicing before upload.
    """

    def __init__(self, call: KernelCall, device=None):
        torch = _torch()
        _abi.require_device()
        # the resident form launches the row kernels on operands that are already what they should be: a call that still
        # carries device-side preprocessing (SP_FLAG_P3_PREP, a CSC matrix1, norms to be built by the library) belongs to
        # _host.run_hip — dropping those options silently would compute something else
        pending = [n for n, v in (("p3_alpha", call.p3_alpha), ("depop_rowsum_p2", call.depop_rowsum_p2),
                                  ("m1_is_m2t", call.m1_is_m2t or None), ("norms_on_device", call.norms_on_device),
                                  ("binary_on_device", call.binary_on_device or None),
                                  ("check_m2_sorted", call.check_m2_sorted or None)) if v is not None]
        if call.col_keep is not None and not call.m2_is_m1t:
            pending.append("col_keep on an explicit matrix2")
        if pending:
            raise ValueError(f"DeviceProblem: the call leaves {', '.join(pending)} to the library's host-mode entry; prepare it without "
                             f"those options (prepare(..., m2_on_device=...) only) or run it through _host.run_hip")
        self.call = call
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=False)  # noqa: E731
        self.t = {}
        for name in ("targets", "m1_data", "m1_indices", "m1_indptr", "m2_data", "m2_indices", "m2_indptr",
                     "Xtversky", "Ytversky", "Xcosine", "Ycosine", "Xdepop", "Ydepop",
                     "filter_m_indptr", "filter_m_indices", "target_col_m_indptr", "target_col_m_indices"):
            arr = getattr(call, name)
            # one device copy per host array: a MATRIX selector that is m1's own pattern (filter_cols = the URM being scored) goes up once
            twin = next((n for n in self.t if self.t[n] is not None and getattr(call, n) is arr), None)
            self.t[name] = self.t[twin] if twin else (up(arr) if arr.size else None)
        self.t["col_keep"] = up(call.col_keep) if call.col_keep is not None and call.col_keep.size else None
        self._ws = None
        self._prep_sig = None      # (flags, tuning) of the run that built the per-call passes now in the workspace

    # ------------------------------------------------------------------
    def _args(self, targets_t, n_targets, cols_t, vals_t, counts_t, rows_t, stream, flags, tuning):
        c = self.call
        a = _abi.SpKnnArgs()
        a.flags = flags | (0 if rows_t is not None else _abi.SP_FLAG_NO_ROWS_OUT) | (_abi.SP_FLAG_M2_IS_M1_T if c.m2_is_m1t else 0)
        a.on_device = 1
        a.device = self.device.index or 0
        a.n_targets, a.n_rows_m1, a.n_rows_m2, a.n_output_cols = n_targets, c.n_rows_m1, c.n_rows_m2, c.n_output_cols
        a.nnz_m1, a.nnz_m2 = int(c.m1_data.shape[0]), int(c.m2_data.shape[0])
        p = lambda t: (t.data_ptr() if t is not None else None)  # noqa: E731
        a.targets = p(targets_t)
        a.col_keep = p(self.t["col_keep"])
        a.m1_data, a.m1_indices, a.m1_indptr = p(self.t["m1_data"]), p(self.t["m1_indices"]), p(self.t["m1_indptr"])
        a.m2_data, a.m2_indices, a.m2_indptr = p(self.t["m2_data"]), p(self.t["m2_indices"]), p(self.t["m2_indptr"])
        a.Xtversky, a.Ytversky = p(self.t["Xtversky"]), p(self.t["Ytversky"])
        a.Xcosine, a.Ycosine = p(self.t["Xcosine"]), p(self.t["Ycosine"])
        a.Xdepop, a.Ydepop = p(self.t["Xdepop"]), p(self.t["Ydepop"])
        a.a1, a.l1, a.l2, a.l3, a.t1, a.t2 = c.a1, c.l1, c.l2, c.l3, c.t1, c.t2
        a.stabilized_shrink, a.bayesian_shrink, a.threshold = c.stabilized_shrink, c.bayesian_shrink, c.threshold
        a.k = c.k
        a.filter_mode, a.target_col_mode = c.filter_mode, c.target_col_mode
        a.filter_m_indptr, a.filter_m_indices = p(self.t["filter_m_indptr"]), p(self.t["filter_m_indices"])
        a.filter_nnz = int(c.filter_m_indices.shape[0])
        a.target_col_m_indptr, a.target_col_m_indices = p(self.t["target_col_m_indptr"]), p(self.t["target_col_m_indices"])
        a.target_col_nnz = int(c.target_col_m_indices.shape[0])
        a.rows, a.cols, a.values, a.out_counts = p(rows_t), p(cols_t), p(vals_t), p(counts_t)
        a.stream = stream
        a.table_slots = int(tuning.get("table_slots", 0))
        a.threads_per_wg = int(tuning.get("threads_per_wg", 0))
        a.num_wgs = int(tuning.get("num_wgs", 0))
        a.load_pct = int(tuning.get("load_pct", 0))
        a.reserved[0] = int(tuning.get("dbg", 0))      # profiling ablations only
        return a

    def alloc_outputs(self, n_targets: Optional[int] = None, with_rows: bool = False):
        torch = _torch()
        n = self.call.n_targets if n_targets is None else int(n_targets)
        k = self.call.k
        cols = torch.empty(n * k, dtype=torch.int32, device=self.device)
        vals = torch.empty(n * k, dtype=torch.float32, device=self.device)
        counts = torch.empty(n, dtype=torch.int32, device=self.device)
        rows = torch.empty(n * k, dtype=torch.int32, device=self.device) if with_rows else None
        return cols, vals, counts, rows

    def run(self, cols, vals, counts, rows=None, targets=None, time_kernel: bool = False,
            static_sched: bool = False, **tuning):
        """Launch on torch's current stream for `targets` (a device int32 tensor; default: the
        problem's own target list).  Asynchronous unless time_kernel=True.  Returns an info dict."""
        torch = _torch()
        targets_t = self.t["targets"] if targets is None else targets
        n = 0 if targets_t is None else int(targets_t.shape[0])
        if n == 0:
            return {"kernel_ms": 0.0, "passes_total": 0}
        flags = (_abi.SP_FLAG_TIME_KERNEL if time_kernel else 0) | (_abi.SP_FLAG_STATIC_SCHED if static_sched else 0)
        if time_kernel and tuning.get("phase_timers", True):
            flags |= _abi.SP_FLAG_PHASE_TIMERS
        if tuning.get("no_sparse_path"):
            flags |= _abi.SP_FLAG_NO_SPARSE_PATH
        if tuning.get("no_fold"):
            flags |= _abi.SP_FLAG_NO_FOLD
        if tuning.get("no_row_order"):
            flags |= _abi.SP_FLAG_NO_ROW_ORDER
        # SP_FLAG_REUSE_M2_PREP only for a run whose layout-relevant flags and tuning are those of the run that built the passes (the
        # call's scalars are this problem's own): folded values, packed terms and window boundaries are laid out per (flags, tuning)
        # — reused under other ones they would be read as something else (the library refuses such a call as well)
        sig = (flags & (_abi.SP_FLAG_NO_FOLD | _abi.SP_FLAG_NO_SPARSE_PATH), int(tuning.get("table_slots", 0)), int(tuning.get("threads_per_wg", 0)),
               int(tuning.get("load_pct", 0)), int(tuning.get("dbg", 0)) & (1024 | 2048 | 4096 | 16384 | 32768 | 65536 | 524288 | 1048576))
        if tuning.get("reuse_m2_prep") and self._ws is not None and self._prep_sig == sig:
            flags |= _abi.SP_FLAG_REUSE_M2_PREP
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            a = self._args(targets_t, n, cols, vals, counts, rows, stream, flags, tuning)
            need = _abi.workspace_bytes(a)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(max(need, 4096), dtype=torch.uint8, device=self.device)
                a.flags &= ~_abi.SP_FLAG_REUSE_M2_PREP       # (a fresh workspace holds nothing to reuse)
            a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
            if not (a.flags & _abi.SP_FLAG_REUSE_M2_PREP):
                self._prep_sig = sig
            _abi.call_knn(a)
        return {"kernel_ms": float(a.kernel_ms), "passes_total": int(a.passes_total), "phase_cycles": [int(x) for x in a.phase_cycles], "num_wgs": int(a.num_wgs_used),
                "sparse_kernel_ms": int(a.reserved[1]) / 1e3, "generic_kernel_ms": int(a.reserved[2]) / 1e3, "transpose_ms": int(a.reserved[3]) / 1e3}
