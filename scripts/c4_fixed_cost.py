import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.distributed import row_work, slice_call
from similaripy_amd.workloads import movielens_like_urm
urm = movielens_like_urm(); m1 = urm.T.tocsr()
call = _host.prepare(m1, k=200, l2=1)
macs = row_work(call)
order = np.argsort(macs)
for name, rows in (("64 lightest rows", order[:64]), ("64 median rows", order[42000:42064]), ("the heaviest row", order[-1:]), ("8 heaviest", order[-8:]), ("first 10554 rows", np.arange(10554))):
    import copy
    c = copy.copy(call); c.targets = np.sort(rows).astype(np.int32)
    prob = DeviceProblem(c); cols, vals, counts, _ = prob.alloc_outputs()
    prob.run(cols, vals, counts); torch.cuda.synchronize()
    i = min((prob.run(cols, vals, counts, time_kernel=True, phase_timers=False) for _ in range(3)), key=lambda d: d["kernel_ms"])
    print(f"{name}: call {i['kernel_ms']:.3f} ms, sparse kernel {i['sparse_kernel_ms']:.3f}, generic kernel {i['generic_kernel_ms']:.3f}, MACs {macs[rows].sum()/1e6:.1f} M", flush=True)
