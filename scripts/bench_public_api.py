"""End-to-end wall clock of the public call (the reference harness's definition: perf_counter around the wrapper), with the
host stages timed apart: prepare (CSR conversion, norm vectors; the transpose too with --host-transpose) | C-ABI call with host buffers (H2D, kernels,
D2H) | output assembly.  Synthetic fixed-degree matrix as bench.py's C2, rows scaled down to keep host memory bounded."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np, scipy.sparse as sp
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import similaripy_amd as sim
from similaripy_amd import _host

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=250_000)
ap.add_argument("--cols", type=int, default=100_000)
ap.add_argument("--nnz-row", type=int, default=64)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--format", default="csr")
ap.add_argument("--host-transpose", action="store_true", help="build m1.T with scipy on the host (the path before SP_FLAG_M2_IS_M1_T)")
a = ap.parse_args()
rng = np.random.default_rng(12345)
cols = rng.integers(0, a.cols, (a.rows, a.nnz_row), dtype=np.int32); cols.sort(1)
data = rng.random(a.rows * a.nnz_row, dtype=np.float32)
m = sp.csr_array((data, cols.ravel(), np.arange(0, a.rows * a.nnz_row + 1, a.nnz_row, dtype=np.int64)), shape=(a.rows, a.cols))
m.sum_duplicates()
sim.cosine(m[:2000], k=10, verbose=False)     # warm-up: library load, device init
t0 = time.perf_counter()
call = _host.prepare(m, None, l2=1.0, c1=0.5, c2=0.5, k=a.k, format_output=a.format, m2_on_device=not a.host_transpose)
t1 = time.perf_counter()
rows, cols_o, vals, counts, info = _host.run_hip(call, time_kernel=True)
t2 = time.perf_counter()
out = _host.finish(call, rows, cols_o, vals, counts, a.format)
t3 = time.perf_counter()
t4 = time.perf_counter()
S = sim.cosine(m, k=a.k, verbose=False, format_output=a.format)
t5 = time.perf_counter()
print(json.dumps({"workload": f"cosine(m) public call, {a.rows}x{a.cols}, nnz/row={a.nnz_row}, k={a.k}, out={a.format}",
                  "prepare_s": round(t1 - t0, 3), "abi_call_host_buffers_s": round(t2 - t1, 3), "kernel_ms": round(info["kernel_ms"], 2), "transpose_ms": round(info["transpose_ms"], 2),
                  "finish_s": round(t3 - t2, 3), "public_call_s": round(t5 - t4, 3), "rows_per_s_end_to_end": round(a.rows / (t5 - t4)), "out_nnz": int(S.nnz)}))
