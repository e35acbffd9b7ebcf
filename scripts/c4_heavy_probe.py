"""configs[3] (rp3beta item-item on the MovieLens-32M-shaped URM, k = 200): the HEAVY rows of the generic kernel (the ones the launch cuts
into column-window pieces: MACs >= 2 x 2^21) timed apart from the light ones, with the in-kernel phase counters of each run.
usage: python scripts/c4_heavy_probe.py [dbg=BITS] [T=table_slots] [nt=threads_per_wg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.distributed import row_work
from similaripy_amd.normalization import normalize
from similaripy_amd.workloads import movielens_like_urm

dbg = 0
tune = {}
for a in sys.argv[1:]:
    if a.startswith("dbg="):
        dbg = int(a[4:])
    if a.startswith("T="):
        tune["table_slots"] = int(a[2:])
    if a.startswith("nt="):
        tune["threads_per_wg"] = int(a[3:])
k, alpha, beta = 200, 0.8, 0.4
urm = movielens_like_urm()
m1 = urm.T.tocsr()
pop_m2 = np.asarray(m1.T.sum(axis=0)).ravel()
a_ = normalize(m1, norm="l1", axis=1); a_.data = np.power(a_.data, np.float32(alpha))
b_ = normalize(m1.T.tocsr(), norm="l1", axis=1); b_.data = np.power(b_.data, np.float32(alpha))
call = _host.prepare(a_, b_, k=k, weight_depop_matrix2=pop_m2, p2=beta, l3=1)
macs = row_work(call)
n1 = np.diff(a_.indptr)
heavy = macs >= 2 * (1 << 21)
print(f"rows {macs.shape[0]}, MACs {macs.sum() / 1e9:.2f} G; heavy rows {int(heavy.sum())}: {macs[heavy].sum() / 1e9:.2f} G MACs, "
      f"{n1[heavy].sum() / 1e6:.2f} M m1 entries (of {n1.sum() / 1e6:.2f} M); MACs per m1 entry heavy {macs[heavy].sum() / n1[heavy].sum():.0f} light {macs[~heavy].sum() / n1[~heavy].sum():.0f}", flush=True)
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
names = ("setup", "segments", "accumulate", "drain", "select", "output", "sweep1", "sweep2")
for label, sel in (("all", np.ones_like(heavy)), ("heavy", heavy), ("light", ~heavy)):
    t = torch.from_numpy(np.nonzero(sel)[0].astype(np.int32)).cuda()
    kw = dict(tune, **(dict(dbg=dbg) if dbg else {}))
    prob.run(cols, vals, counts, targets=t, **kw); torch.cuda.synchronize()
    ms = min(prob.run(cols, vals, counts, targets=t, time_kernel=True, phase_timers=False, **kw)["kernel_ms"] for _ in range(3))
    info = prob.run(cols, vals, counts, targets=t, time_kernel=True, **kw)
    cyc = info.get("phase_cycles", [0] * 12)
    tot = float(sum(cyc[:8])) or 1.0
    print(f"{label:6s} rows {int(sel.sum()):6d}  kernel {ms:7.2f} ms   " + "  ".join(f"{n} {c / tot:.3f}" for n, c in zip(names, cyc[:8])) +
          f"   windows {cyc[11] & 0xFFFFFFFF}  wgs {info.get('num_wgs')}   MACs/clk/CU {macs[sel].sum() / (ms * 1e-3 * 2.4e9 * 256):.2f}", flush=True)
