import numpy as np, scipy.sparse as sp, sys, collections
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
n_cols = 50000
rng = np.random.default_rng(1)
def run(shared_cols, seglen=600, nseg=2, **kw):
    rows_m2 = []
    for s in range(nseg):
        cols = set(rng.choice(np.arange(100, n_cols), size=seglen, replace=False).tolist()) | set(shared_cols)
        rows_m2.append(sorted(cols))
    indptr = np.cumsum([0] + [len(r) for r in rows_m2]).astype(np.int32)
    indices = np.concatenate(rows_m2).astype(np.int32)
    data = rng.random(indices.shape[0], dtype=np.float32) + 0.5
    m2 = sp.csr_array((data, indices, indptr), shape=(nseg, n_cols))
    m1 = sp.csr_array((np.linspace(2.0, 1.0, nseg).astype(np.float32), np.arange(nseg, dtype=np.int32), np.array([0, nseg], dtype=np.int32)), shape=(1, nseg))
    call = _host.prepare(m1, m2, k=5000)
    r = _host.run_hip(call, time_kernel=True, **kw)
    n = r[3][0]
    gc = r[1][:n]
    cnt = collections.Counter(gc.tolist())
    missed = [c for c, k_ in cnt.items() if k_ > 1]
    allc = collections.Counter(indices.tolist())
    coll = sorted(c for c, k_ in allc.items() if k_ > 1)
    total = indices.shape[0]
    nw = kw.get("threads_per_wg", 512) // 64
    chunk = -(-total // (nw * 64)) * 64
    def where(c):
        flat = np.flatnonzero(indices == c)
        return [(int(f), int(f // chunk), int((f % chunk) % 64), int((f % chunk) // 64)) for f in flat]   # flat, wave, lane, j
    print("total", total, "chunk", chunk, "collision cols", len(coll), "missed", len(missed))
    for c in coll:
        print("   col", c, "MISSED" if c in missed else "ok    ", "(flat, wave, lane, j):", where(c))
run([5])
run([5], threads_per_wg=1024)
