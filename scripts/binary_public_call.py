"""Wall clock of public calls with binary=True at the C2 size (SP_FLAG_BINARY: the ones are written on the device)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import similaripy_amd as sim
from similaripy_amd import workloads
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
# (C2's fixed-degree rows give every row the same norm: under `binary` nearly all candidates of a row tie — the C4 shape has real norms)
m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345) if which == "c2" else workloads.movielens_like_urm().T.tocsr()
sim.cosine(m[:2000], k=10, verbose=False)
for name, f in (("cosine", lambda: sim.cosine(m, k=100, verbose=False, format_output="csr")),
                ("jaccard", lambda: sim.jaccard(m, k=100, verbose=False, format_output="csr")),
                ("cosine binary", lambda: sim.cosine(m, k=100, binary=True, verbose=False, format_output="csr")),
                ("jaccard binary", lambda: sim.jaccard(m, k=100, binary=True, verbose=False, format_output="csr")),
                ("tversky binary", lambda: sim.tversky(m, alpha=0.4, beta=0.6, k=100, binary=True, verbose=False, format_output="csr"))):
    f()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0); del r
    print(f"{name:16s} {min(ts):.3f} s", flush=True)
