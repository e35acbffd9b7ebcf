"""C5-shaped synthetic (BASELINE configs[4]): user scoring  dot_product(urm, W.T, k=100, filter_cols=urm)
with W = cosine(urm_small.T, k=100) — one GPU's slice of the 10M-user job (default 1M users x 100k items, 64 nnz/row).
Kernel-scope timing + parity on a sample of rows (MATRIX filter selector, general sparse kernel)."""
import sys, time, json, copy
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import torch
import similaripy_amd as sim
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from oracle import splus_oracle as so
from bench import fixed_degree_csr

n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
# optional tuning sweep: "threads_per_wg=512,table_slots=8192;threads_per_wg=512,table_slots=4096"
sweeps = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in grp.split(",")) for grp in sys.argv[2].split(";")] if len(sys.argv) > 2 else []
n_items, k = 100_000, 100
urm = fixed_degree_csr(n_users, n_items, 64, 12345)
t0 = time.perf_counter()
small = urm[: min(n_users, 200_000)]
W = sim.cosine(small.T.tocsr(), k=100, verbose=False, format_output="csr")      # item-item model from a user subsample
print(f"W {W.shape} nnz {W.nnz} built in {time.perf_counter() - t0:.1f}s (public API, host prep included)", flush=True)
call = _host.prepare(urm, W.T.tocsr(), k=k, filter_cols=urm)
nnz2 = np.diff(call.m2_indptr).astype(np.int64)
per = nnz2[call.m1_indices]; cs = np.concatenate(([0], np.cumsum(per)))
macs = cs[call.m1_indptr[1:]] - cs[call.m1_indptr[:-1]]
prob = DeviceProblem(call)
cols, vals, counts, _ = prob.alloc_outputs()
prob.run(cols, vals, counts); torch.cuda.synchronize()
info = prob.run(cols, vals, counts, time_kernel=True, phase_timers=False)
ms = info["kernel_ms"]
nbytes = 16 * call.m1_data.shape[0] + 8 * int(macs.sum()) + 8 * k * call.n_targets
info2 = prob.run(cols, vals, counts, time_kernel=True)
ph = info2["phase_cycles"]
print(json.dumps({"workload": "dot_product + filter_cols (C5 slice)", "rows": call.n_targets, "macs_per_row": float(macs.mean()),
                  "call_ms": ms, "sparse_kernel_ms": info["sparse_kernel_ms"], "generic_kernel_ms": info["generic_kernel_ms"],
                  "rows_per_s": call.n_targets / ms * 1e3, "algorithmic_GBps": nbytes / ms / 1e6,
                  "rows_sparse": ph[9], "rows_given_up": ph[10], "generic_windows": ph[11]}), flush=True)
sample = np.sort(np.random.default_rng(1).choice(call.n_targets, 200, replace=False)).astype(np.int32)
c2 = copy.copy(call); c2.targets = sample
want = so.canonical(*so.run_kernel(c2, "port"), sample, k)
hc, hv, hn = cols.cpu().numpy(), vals.cpu().numpy(), counts.cpu().numpy()
got = []
for t in sample:
    n = hn[t]; cc = hc[t*k:t*k+n]; vv = hv[t*k:t*k+n]; o = np.argsort(cc); got.append((cc[o], vv[o]))
ties = so.compare_topk(got, want, k, rtol=1e-5, atol=1e-7, what="c5")
# the filter must hold: no recommended item is one the user already has
for t in sample[:50]:
    assert not np.intersect1d(hc[t*k:t*k+hn[t]], urm.indices[urm.indptr[t]:urm.indptr[t+1]]).size
print(f"   parity OK on {len(sample)} rows (boundary ties {ties}); filter respected", flush=True)
for tun in sweeps:
    prob.run(cols, vals, counts, **tun); torch.cuda.synchronize()
    i1 = prob.run(cols, vals, counts, time_kernel=True, phase_timers=False, **tun)
    i2 = prob.run(cols, vals, counts, time_kernel=True, **tun)
    print(f"   tuning {tun}: {i1['kernel_ms']:.1f} ms, workgroups {i1['num_wgs']}, rows sparse / given up {i2['phase_cycles'][9]} / {i2['phase_cycles'][10]}", flush=True)
names = ("setup", "segments", "accumulate", "drain", "select", "output", "sweep1", "sweep2", "csdrain")
tot = float(sum(ph[:9])); per_row = tot / call.n_targets
print("   cycles/row %.0f: " % per_row + "  ".join(f"{n}={c / call.n_targets:.0f}" for n, c in zip(names, ph[:9])), flush=True)
