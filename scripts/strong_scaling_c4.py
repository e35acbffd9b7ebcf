"""What row sharding gives on the MovieLens-32M-shaped item-item call (BASELINE configs[3]) — on ONE GPU: the slices
`distributed.partition_targets` cuts for N ranks are run one after the other, the slowest one is the N-GPU step
(the gather of 84 k x k results is small).  usage: python scripts/strong_scaling_c4.py [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.distributed import partition_targets, row_work, slice_call
from similaripy_amd.workloads import movielens_like_urm

k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
urm = movielens_like_urm(); m1 = urm.T.tocsr()
call = _host.prepare(m1, k=k, l2=1)
work = row_work(call)
base = None
for world in (1, 2, 4, 8):
    b = partition_targets(work, world)
    res = {}
    for tag, dbg in (("pieces", 0), ("whole rows", 4096)):
        times = []
        for r in range(world):
            sub = slice_call(call, int(b[r]), int(b[r + 1]), compact=True)
            prob = DeviceProblem(sub); cols, vals, counts, _ = prob.alloc_outputs()
            prob.run(cols, vals, counts, dbg=dbg); torch.cuda.synchronize()
            times.append(min(prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, dbg=dbg)["kernel_ms"] for _ in range(2)))
            del prob
        res[tag] = max(times)
    if base is None:
        base = dict(res)
    print(f"N={world}: slowest slice {res['pieces']:.1f} ms (x{base['pieces'] / res['pieces']:.2f}); heavy rows not cut: {res['whole rows']:.1f} ms (x{base['whole rows'] / res['whole rows']:.2f})", flush=True)
