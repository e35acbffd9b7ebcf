"""Phase split of the generic row kernel on the MovieLens-32M-shaped item-item call (BASELINE configs[3] shape).
usage: python scripts/phases_c4.py [k] ["threads_per_wg=512,table_slots=8192;..."]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import movielens_like_urm

k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sweeps = [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in grp.split(',')) for grp in sys.argv[2].split(';')] if len(sys.argv) > 2 else []
urm = movielens_like_urm()
m1 = urm.T.tocsr()
call = _host.prepare(m1, k=k, l2=1)
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
names = ["setup", "segments", "accumulate", "judge", "select", "output", "sweep1", "sweep2", "-", "rows_sparse", "rows_to_generic", "windows"]
for tun in [{}] + sweeps:
    prob.run(cols, vals, counts, **tun); torch.cuda.synchronize()
    i0 = prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, **tun)
    info = prob.run(cols, vals, counts, time_kernel=True, **tun)
    ph = info["phase_cycles"]
    tot = sum(ph[:6])
    print(f"k={k} {tun}: kernel {i0['kernel_ms']:.2f} ms (with phase timers {info['kernel_ms']:.2f}), workgroups {info['num_wgs']}, windows {ph[11]}, repeated sweeps {ph[8]}, passes {info['passes_total']}")
    print("   " + "  ".join(f"{n} {100.0 * c / tot:.1f}%" for n, c in zip(names[:8], ph[:8]) if c))
