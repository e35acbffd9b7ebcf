"""C2-shaped problem (1M x 100k, 64 per row, cosine k=100; `c3` = the s_plus hybrid of configs[2]) over the first N target rows,
kernel-scope: how long the row kernels took, where the cycles of a row went (in-kernel phase timers), parity on a sample.
`python scripts/c2_phases.py N [c3|jaccard] [binary] [static] [dbg=BITS] [wgs=N]`  (dbg=524288: the two-per-CU shape off)"""
import sys, json, copy
import numpy as np
sys.path.insert(0, '.')
import torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import fixed_degree_csr
from oracle import splus_oracle as so

n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tun = dict(dbg=next((int(a[4:]) for a in sys.argv if a.startswith("dbg=")), 0))
if any(a.startswith("wgs=") for a in sys.argv):      # persistent workgroups of the row kernels (default: what fills the CUs)
    tun["num_wgs"] = next(int(a[4:]) for a in sys.argv if a.startswith("wgs="))
kw = dict(l1=0.5, l2=0.5, stabilized_shrink=10.0) if "c3" in sys.argv else dict(l1=1, t1=1, t2=1) if "jaccard" in sys.argv else dict(l2=1, c1=0.5, c2=0.5)
if "binary" in sys.argv:      # (fixed-degree rows of ones: every row has the same norm, nearly every candidate of a row ties)
    kw["binary"] = True
thr = next((float(a[4:]) for a in sys.argv if a.startswith("thr=")), None)
if thr is not None:
    kw["threshold"] = thr
static = "static" in sys.argv
m = fixed_degree_csr(1_000_000, 100_000, 64, 12345)
k = 100
call = _host.prepare(m, k=k, target_rows=np.arange(n_t), **kw)
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
print("launch", n_t, tun, flush=True)
prob.run(cols, vals, counts, static_sched=static, **tun); torch.cuda.synchronize()
print("first pass done", flush=True)
i1 = prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, static_sched=static, **tun)
i2 = prob.run(cols, vals, counts, time_kernel=True, static_sched=static, **tun)
ph = i2["phase_cycles"]
names = ("setup", "segments", "accumulate", "drain", "select", "output", "sweep1", "sweep2", "csdrain")
print(json.dumps({"rows": n_t, "call_ms": i1["kernel_ms"], "sparse_ms": i1["sparse_kernel_ms"], "generic_ms": i1["generic_kernel_ms"],
                  "rows_sparse": ph[9], "given_up": ph[10] & 0xFFFFFFFF,
                  "given_up_why": dict(zip(("items_or_row", "collision_set", "U_full", "member_pool_full"), [(ph[10] >> s) & 0xFF for s in (32, 40, 48, 56)]))}), flush=True)
tot = float(sum(ph[:9]))
print("   cycles/row %.0f: " % (tot / n_t) + "  ".join(f"{n}={c / n_t:.0f}" for n, c in zip(names, ph[:9])), flush=True)
if ph[8] & 2:      # the bounded variant ran: its counters (slot 11: selections << 32 | entries through the exact pass)
    print(f"   bounded variant: {(ph[11] & 0xFFFFFFFF) / n_t:.0f} entries through the exact pass per row, {(ph[11] >> 32) / n_t:.2f} selections per row", flush=True)
if tun["dbg"] & ~524288:
    sys.exit(0)
sample = np.sort(np.random.default_rng(1).choice(n_t, min(n_t, 150), replace=False)).astype(np.int32)
c2 = copy.copy(call); c2.targets = sample
want = so.canonical(*so.run_kernel(c2, "port"), sample, k)
hc, hv, hn = cols.cpu().numpy(), vals.cpu().numpy(), counts.cpu().numpy()
got = []
for t in sample:
    n = hn[t]; cc = hc[t*k:t*k+n]; vv = hv[t*k:t*k+n]; o = np.argsort(cc); got.append((cc[o], vv[o]))
ties = so.compare_topk(got, want, k, rtol=1e-5, atol=1e-7, what="probe")
print(f"   parity OK on {len(sample)} rows (boundary ties {ties})", flush=True)
