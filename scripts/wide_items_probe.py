import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
rng = np.random.default_rng(5)
n_users, n_items = 400000, 1000000
urm = sp.random_array((n_users, n_items), density=64 / n_items, format="csr", dtype=np.float32, random_state=rng)
wt = sp.random_array((n_items, n_items), density=100 / n_items, format="csr", dtype=np.float32, random_state=rng)
call = _host.prepare(urm, wt, k=100, filter_cols=urm)
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
for dbg in (0, 16384):
    prob.run(cols, vals, counts, dbg=dbg); torch.cuda.synchronize()
    i = [prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, dbg=dbg) for _ in range(3)]
    print("dbg", dbg, "kernel_ms", min(x["kernel_ms"] for x in i), "sparse_ms", min(x["sparse_kernel_ms"] for x in i), "generic", i[0]["generic_kernel_ms"], "wgs", i[0]["num_wgs"], flush=True)
