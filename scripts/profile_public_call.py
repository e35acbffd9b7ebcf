"""Where the wall clock of the public calls goes (host preparation / C-ABI call with host buffers / output assembly).
usage: python scripts/profile_public_call.py [c2|c4|c5]"""
import sys, time, cProfile, pstats, io
sys.path.insert(0, '.')
import numpy as np
import similaripy_amd as sim
from similaripy_amd import workloads

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
    calls = [("cosine csr", lambda: sim.cosine(m, k=100, verbose=False, format_output="csr")),
             ("cosine coo", lambda: sim.cosine(m, k=100, verbose=False, format_output="coo")),
             # ARRAY selector: a tenth of the columns dropped while m2 = m1^T is built on the device (sp_knn_args.col_keep)
             ("cosine csr filter_cols=list", lambda: sim.cosine(m, k=100, verbose=False, format_output="csr", filter_cols=list(range(0, 1_000_000, 10))))]
elif which == "c5":
    m = workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)
    Wt = sim.cosine(m[:200_000].T.tocsr(), k=100, verbose=False, format_output="csr").T.tocsr()
    calls = [("dot_product(urm, W.T, filter_cols=urm)", lambda: sim.dot_product(m, Wt, k=100, filter_cols=m, verbose=False, format_output="csr"))]
else:
    m = workloads.movielens_like_urm().T.tocsr()
    calls = [("cosine", lambda: sim.cosine(m, k=200, verbose=False, format_output="csr")),
             ("p3alpha", lambda: sim.p3alpha(m, alpha=0.8, k=200, verbose=False, format_output="csr")),
             ("rp3beta", lambda: sim.rp3beta(m, alpha=0.8, beta=0.4, k=200, verbose=False, format_output="csr"))]
sim.cosine(m[:2000], k=10, verbose=False)
for name, f in calls:
    f()
    t0 = time.perf_counter(); res = f(); t1 = time.perf_counter()      # (the result is released OUTSIDE the timed region: unmapping 0.8 GB of touched pages takes 30-40 ms)
    del res
    pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumtime").print_stats(18)
    print(f"== {which} {name}: {t1 - t0:.3f} s")
    for l in st.getvalue().splitlines():
        if any(s in l for s in ("similaripy_amd", "scipy", "numpy", "method", "built-in")):
            print("   ", l.strip()[:160])
