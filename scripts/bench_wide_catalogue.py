"""User scoring over a catalogue of MORE than 2^18 items (light rows, aliasing bitmap): the auto-tuned 256-thread shape of the
sparse kernel against the 1024-thread one.  dot_product(urm, W.T, k=100), 500k users x 100k "source" items, 1M scored items."""
import sys, json
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from bench import fixed_degree_csr
n_users, n_src, n_items = 500_000, 100_000, 1_000_000
urm = fixed_degree_csr(n_users, n_src, 64, 12345)
rng = np.random.default_rng(7)
cols = rng.integers(0, n_items, (n_src, 100), dtype=np.int32); cols.sort(1)
W = sp.csr_array((rng.random(n_src * 100, dtype=np.float32), cols.ravel(), np.arange(0, n_src * 100 + 1, 100, dtype=np.int32)), shape=(n_src, n_items)); W.sum_duplicates()
call = _host.prepare(urm, W, k=100)
prob = DeviceProblem(call)
cols_o, vals, counts, _ = prob.alloc_outputs()
for tun in ({}, dict(threads_per_wg=1024, table_slots=16384)):
    prob.run(cols_o, vals, counts, **tun); torch.cuda.synchronize()
    i1 = prob.run(cols_o, vals, counts, time_kernel=True, phase_timers=False, **tun)
    i2 = prob.run(cols_o, vals, counts, time_kernel=True, **tun)
    print(json.dumps({"tuning": tun or "auto", "kernel_ms": round(i1["kernel_ms"], 2), "workgroups": i1["num_wgs"], "rows_per_s": round(n_users / i1["kernel_ms"] * 1e3),
                      "rows_sparse": i2["phase_cycles"][9], "rows_given_up": i2["phase_cycles"][10]}), flush=True)
