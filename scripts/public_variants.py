"""Wall clock of the public wrappers and their argument variants at the C2 size (1 M x 100 k, 64 per row, k = 100, CSR out):
which variant pays host time on top of the ~0.13 s of the plain cosine call?   python scripts/public_variants.py [only-substring]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp
import similaripy_amd as sim
from similaripy_amd import workloads
m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
mc = m.tocsc()
mT = m.T.tocsr()
rng = np.random.default_rng(0)
P1 = rng.random(1_000_000).astype(np.float32) + 0.5
F = sp.random_array((1_000_000, 1_000_000), density=5e-6, format="csr", dtype=np.float32, random_state=rng)
kw = dict(k=100, verbose=False, format_output="csr")
calls = {
 "cosine": lambda: sim.cosine(m, **kw),
 "cosine csc": lambda: sim.cosine(mc, **kw),
 "cosine coo out": lambda: sim.cosine(m, k=100, verbose=False, format_output="coo"),
 "cosine explicit m2": lambda: sim.cosine(m, mT, **kw),
 "cosine explicit m2 binary": lambda: sim.cosine(m, mT, binary=True, **kw),
 "cosine shrink": lambda: sim.cosine(m, shrink=10, **kw),
 "cosine threshold": lambda: sim.cosine(m, threshold=0.05, **kw),
 "asymmetric_cosine": lambda: sim.asymmetric_cosine(m, alpha=0.3, **kw),
 "jaccard": lambda: sim.jaccard(m, **kw),
 "dice": lambda: sim.dice(m, **kw),
 "tversky": lambda: sim.tversky(m, alpha=0.4, beta=0.6, **kw),
 "dot_product": lambda: sim.dot_product(m, **kw),
 "p3alpha": lambda: sim.p3alpha(m, alpha=0.8, **kw),
 "rp3beta": lambda: sim.rp3beta(m, alpha=0.8, beta=0.4, **kw),
 "s_plus l3=0": lambda: sim.s_plus(m, l1=0.5, l2=0.5, **kw),
 "s_plus pop sum": lambda: sim.s_plus(m, l1=0.5, l2=0.5, l3=0.2, pop1='sum', pop2='sum', beta1=0.5, beta2=0.5, **kw),
 "s_plus pop arrays": lambda: sim.s_plus(m, l1=0.5, l2=0.5, l3=0.2, pop1=P1, pop2=P1, beta1=0.5, beta2=0.5, **kw),
 "cosine target_rows half": lambda: sim.cosine(m, target_rows=np.arange(0, 1_000_000, 2), **kw),
 "cosine target_rows shuffled": lambda: sim.cosine(m, target_rows=rng.permutation(1_000_000)[:500_000], **kw),
 "cosine filter matrix": lambda: sim.cosine(m, filter_cols=F, **kw),
 "cosine target matrix": lambda: sim.cosine(m, target_cols=F, **kw),
 "cosine filter list": lambda: sim.cosine(m, filter_cols=list(range(0, 1_000_000, 10)), **kw),
 "cosine target array": lambda: sim.cosine(m, target_cols=np.arange(0, 1_000_000, 2), **kw),
}
only = sys.argv[1] if len(sys.argv) > 1 else ""
profile = "profile" in sys.argv[2:]      # python scripts/public_variants.py SUBSTRING profile: cProfile of the matching calls
sim.cosine(m[:2000], k=10, verbose=False)
for name, f in calls.items():
    if only and only not in name:
        continue
    f()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0); nnz = r.nnz; del r
    print(f"{name:30s} {min(ts):7.3f} s   nnz {nnz}", flush=True)
    if profile:
        import cProfile, pstats, io
        pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(10)
        for l in st.getvalue().splitlines():
            if any(x in l for x in ("similaripy_amd", "scipy", "numpy", "method", "built-in")):
                print("      ", l.strip()[:170])
