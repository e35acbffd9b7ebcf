"""General epilogues (bounded variant) on BINARY data: fixed-degree rows (every single product of a row ties exactly) against Poisson degrees.
`python scripts/binary_general_probe.py`"""
import sys
import numpy as np
sys.path.insert(0, '.')
import scipy.sparse as sp
import torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import fixed_degree_csr

def run(m, name, **kw):
    t = np.arange(0, 200_000, dtype=np.int32)
    call = _host.prepare(m, k=100, target_rows=t, binary=True, **kw)
    prob = DeviceProblem(call)
    outs = prob.alloc_outputs()[:3]
    for tun, tag in (({}, "chosen "), ({"dbg": 524288}, "classic")):
        prob.run(*outs, **tun); torch.cuda.synchronize()
        i = prob.run(*outs, time_kernel=True, **tun)
        pc = i["phase_cycles"]
        why = dict(zip(("items", "cs", "U", "pool"), [(pc[10] >> s_) & 0xFF for s_ in (32, 40, 48, 56)]))
        print(f"{name:40s} {tag} sparse {i['sparse_kernel_ms']:7.2f} ms  generic {i['generic_kernel_ms']:7.2f} ms  rows on the sparse kernel {pc[9]}  handed over {pc[10] & 0xFFFFFFFF} (mod 256: {why})  stages taken back (mod 2048) {pc[11] >> 53}", flush=True)

fx = fixed_degree_csr(1_000_000, 100_000, 64, 12345)
po = sp.random_array((1_000_000, 100_000), density=64 / 100_000, format="csr", dtype=np.float32, random_state=np.random.default_rng(5))
# skewed column popularity (Zipf-ish), Poisson row degrees
rng = np.random.default_rng(6)
cols = (rng.pareto(1.5, size=po.nnz) * 2000).astype(np.int64) % 100_000
sk = sp.csr_array((np.ones(po.nnz, np.float32), cols.astype(np.int32), po.indptr), shape=po.shape); sk.sum_duplicates(); sk.data[:] = 1
for nm, m in (("fixed degree 64", fx), ("Poisson degrees ~64", po)):
    run(m, nm + ", cosine", l2=1)
    run(m, nm + ", jaccard", l1=1, t1=1, t2=1)
    run(m, nm + ", cosine shrink 10", l2=1, stabilized_shrink=10.0)
