"""C4-shaped synthetic, wall clock of the PUBLIC calls (item-item p3alpha / rp3beta / cosine on URM.T, k=200) with the
host stages of the p3 wrappers timed apart.  Full MovieLens-32M shape by default (scale 1.0)."""
import sys, time, json, cProfile, pstats, io
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
import similaripy_amd as sim
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
def make_urm(n_users, n_items, nnz, seed=0):
    rng = np.random.default_rng(seed)
    act = rng.lognormal(mean=0.0, sigma=1.0, size=n_users); act = act / act.sum()
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.9; pop = pop / pop.sum()
    u = rng.choice(n_users, size=nnz, p=act).astype(np.int32)
    i = rng.choice(n_items, size=nnz, p=pop).astype(np.int32)
    r = (rng.integers(1, 11, size=nnz) * 0.5).astype(np.float32)
    m = sp.csr_array((r, (u, i)), shape=(n_users, n_items)); m.sum_duplicates()
    return m
n_users, n_items, nnz = int(200948 * scale), int(84432 * scale), int(32_000_204 * scale)
urm = make_urm(n_users, n_items, nnz)
m1 = urm.T.tocsr()
print(f"URM {urm.shape} nnz {urm.nnz}; item-item on m1 {m1.shape}", flush=True)
sim.cosine(m1[:500], k=10, verbose=False)
for name, f in (("cosine", lambda: sim.cosine(m1, k=200, verbose=False)), ("p3alpha", lambda: sim.p3alpha(m1, alpha=0.8, k=200, verbose=False)),
                ("rp3beta", lambda: sim.rp3beta(m1, alpha=0.8, beta=0.4, k=200, verbose=False))):
    t0 = time.perf_counter(); S = f(); t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumtime").print_stats(14)
    top = [l.strip()[:150] for l in st.getvalue().splitlines() if ("similaripy_amd" in l or "scipy" in l or "numpy" in l or "method" in l)][:12]
    print(json.dumps({"workload": name + " public call, item-item k=200", "rows": m1.shape[0], "wall_s": round(t1 - t0, 3), "rows_per_s": round(m1.shape[0] / (t1 - t0)), "out_nnz": int(S.nnz)}))
    for l in top: print("     ", l)
