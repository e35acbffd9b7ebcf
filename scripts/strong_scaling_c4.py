"""What row sharding gives on the MovieLens-32M-shaped item-item call (BASELINE configs[3]) — on ONE GPU: the slices
`distributed.partition_targets` cuts for N ranks are run one after the other, the slowest one is the N-GPU step
(the gather of 84 k x k results is small).  Also fits the per-row cost model the partition uses (distributed.row_cost):
slice time ~ a * MACs + b * rows + c * heavy-row pieces, least squares over all the slices timed.
usage: python scripts/strong_scaling_c4.py [k] [macs|cost]      (what the partition balances; default: cost)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.distributed import partition_targets, row_work, row_cost, slice_call
from similaripy_amd.workloads import movielens_like_urm

k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "cost"
urm = movielens_like_urm(); m1 = urm.T.tocsr()
call = _host.prepare(m1, k=k, l2=1)
macs = row_work(call)
work = macs if mode == "macs" else row_cost(call)
pieces_of = np.where(macs >= 2 * (1 << 21), np.minimum(np.ceil(macs / float(1 << 21)), 32), 0)      # (the splitter's rule at N = 1, sp_row_desc_kernel)
print(f"partition balances: {mode}; rows {macs.shape[0]}, MACs {macs.sum() / 1e9:.2f} G, rows cut into pieces {int((pieces_of > 0).sum())}", flush=True)
base = None
obs = []
for world in (1, 2, 4, 8):
    b = partition_targets(work, world)
    times = []
    for r in range(world):
        lo, hi = int(b[r]), int(b[r + 1])
        sub = slice_call(call, lo, hi, compact=True)
        prob = DeviceProblem(sub); cols, vals, counts, _ = prob.alloc_outputs()
        prob.run(cols, vals, counts); torch.cuda.synchronize()
        t = min(prob.run(cols, vals, counts, time_kernel=True, phase_timers=False)["kernel_ms"] for _ in range(3))
        times.append(t)
        obs.append((float(macs[lo:hi].sum()), float(hi - lo), float(pieces_of[lo:hi].sum()), t))
        del prob
    if base is None:
        base = max(times)
    print(f"N={world}: slowest slice {max(times):.2f} ms (x{base / max(times):.2f}); slices " + " ".join(f"{t:.2f}" for t in times), flush=True)
    if world == 8:
        print("   N=8 slices (GMACs, k rows, pieces): " + "  ".join(f"({o[0] / 1e9:.2f}, {o[1] / 1e3:.1f}, {int(o[2])})" for o in obs[-8:]), flush=True)
A = np.array([(o[0] / 1e9, o[1] / 1e3, o[2], 1.0) for o in obs]); y = np.array([o[3] for o in obs])
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print(f"fit: ms = {coef[0]:.3f} * GMACs + {coef[1]:.4f} * krows + {coef[2]:.4f} * pieces + {coef[3]:.3f};  residuals (ms): "
      + " ".join(f"{r:+.2f}" for r in (A @ coef - y)), flush=True)
