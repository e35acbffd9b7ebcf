"""What the split-phase gather's sub-launches cost in compute: ONE rank's slice of the C2 job at N = 8 (125 k rows), run as 1 / 2 / 4 / 8
sub-launches on a resident problem whose passes over m2 are kept (persist_prep) — ms per step without any gather.
    python scripts/sublaunch_cost.py [rows_of_the_slice]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import fixed_degree_csr

n_slice = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
m = fixed_degree_csr(1_000_000, 100_000, 64, 12345)
call = _host.prepare(m, k=100, l2=1, c1=0.5, c2=0.5, target_rows=np.arange(n_slice))
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
k = call.k
tg = prob.t["targets"]
prob.run(cols, vals, counts)
torch.cuda.synchronize()
for phases in (1, 2, 4, 8):
    n_sub = -(-n_slice // phases)
    def step():
        for j in range(phases):
            a, b = j * n_sub, min(n_slice, (j + 1) * n_sub)
            prob.run(cols[a * k: b * k], vals[a * k: b * k], counts[a:b], targets=tg[a:b], reuse_m2_prep=True)
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    host_ms = (time.perf_counter() - t0) * 100
    torch.cuda.synchronize()
    print(f"phases {phases}: {e0.elapsed_time(e1) / 10:.3f} ms per step on the device ({host_ms:.3f} ms of host time to enqueue it)", flush=True)
