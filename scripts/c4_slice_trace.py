"""One N = 8 slice of the MovieLens-32M-shaped item-item call (the slowest of round 3's experiment: slice 5), run three times — under
`rocprofv3 --kernel-trace` this lists every launch of a slice with its duration (where does the per-slice constant live?).
usage: rocprofv3 --kernel-trace --output-format csv -d /tmp/c4tr -o t -- python scripts/c4_slice_trace.py [slice] [world]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.distributed import partition_targets, row_cost, slice_call
from similaripy_amd.workloads import movielens_like_urm

r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
urm = movielens_like_urm(); m1 = urm.T.tocsr()
call = _host.prepare(m1, k=200, l2=1)
b = partition_targets(row_cost(call), world)
sub = slice_call(call, int(b[r]), int(b[r + 1]), compact=True)
prob = DeviceProblem(sub); cols, vals, counts, _ = prob.alloc_outputs()
for _ in range(3):
    prob.run(cols, vals, counts); torch.cuda.synchronize()
info = prob.run(cols, vals, counts, time_kernel=True)
print(f"slice {r} of {world}: rows {sub.n_targets}, call {info['kernel_ms']:.2f} ms, generic {info['generic_kernel_ms']:.2f} ms, windows {info['passes_total']}", flush=True)
pc = info["phase_cycles"]
tot = float(sum(pc[:8])) or 1.0
print("phase share:", " ".join(f"{n}={c / tot:.3f}" for n, c in zip(("setup", "segments", "accumulate", "drain", "select", "output", "s1", "s2"), pc[:8])), "cycles/wg", tot / max(1, info["num_wgs"]))
