import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, copy
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import movielens_like_urm
urm = movielens_like_urm(); m1 = urm.T.tocsr()
call = _host.prepare(m1, k=200, l2=1)
deg = np.diff(m1.indptr); order = np.argsort(-deg)
for n in (1, 8, 64, 256):
    c = copy.copy(call); c.targets = np.sort(order[:n]).astype(np.int32)
    prob = DeviceProblem(c); cols, vals, counts, _ = prob.alloc_outputs()
    prob.run(cols, vals, counts); torch.cuda.synchronize()
    info = prob.run(cols, vals, counts, time_kernel=True, phase_timers=False)
    i2 = prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, dbg=4096)
    print(f"{n} heaviest rows (n1 {deg[order[0]]}..{deg[order[n-1]]}): call {info['kernel_ms']:.2f} ms, generic row kernel {info['generic_kernel_ms']:.2f} ms; without pieces {i2['generic_kernel_ms']:.2f} ms")
