"""General epilogues on the MovieLens-32M-shaped item-item call (generic kernel, dense windows): jaccard and cosine with a shrink, k = 200;
the list-based drain (round 5) against the old sweep (ablation bit 262144).  python scripts/c4_general_epilogues.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, copy
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import movielens_like_urm
from oracle import splus_oracle as so
m1 = movielens_like_urm().T.tocsr()
k = 200
for name, kw in (("jaccard", dict(l1=1)), ("cosine shrink 10", dict(l2=1, stabilized_shrink=10)), ("s_plus l1=l2=.5 shrink 10", dict(l1=0.5, l2=0.5, stabilized_shrink=10))):
    call = _host.prepare(m1, k=k, **kw)
    prob = DeviceProblem(call)
    cols, vals, counts, _ = prob.alloc_outputs()
    for dbg in (0, 262144):
        prob.run(cols, vals, counts, dbg=dbg); torch.cuda.synchronize()
        ms = min(prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, dbg=dbg)["kernel_ms"] for _ in range(2))
        print(f"{name:28s} dbg={dbg:6d}  kernel {ms:7.2f} ms", flush=True)
    prob.run(cols, vals, counts); torch.cuda.synchronize()
    sample = np.sort(np.random.default_rng(1).choice(call.n_targets, 60, replace=False)).astype(np.int32)
    c2 = copy.copy(call); c2.targets = sample
    want = so.canonical(*so.run_kernel(c2, "port"), sample, k)
    hc, hv, hn = cols.cpu().numpy(), vals.cpu().numpy(), counts.cpu().numpy()
    got = []
    for t in sample:
        n = hn[t]; cc = hc[t*k:t*k+n]; vv = hv[t*k:t*k+n]; o = np.argsort(cc); got.append((cc[o], vv[o]))
    print("   parity OK, ties", so.compare_topk(got, want, k, rtol=1e-3, atol=1e-9, what=name), flush=True)
