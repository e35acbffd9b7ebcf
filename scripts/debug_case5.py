import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
n_cols = 50000
rng = np.random.default_rng(1)
def run(shared_cols, seglen=600, nseg=2):
    # m1: 1 row with nseg nnz; m2: nseg rows with `seglen` random distinct columns each + the shared ones
    rows_m2 = []
    for s in range(nseg):
        cols = set(rng.choice(np.arange(100, n_cols), size=seglen, replace=False).tolist()) | set(shared_cols)
        rows_m2.append(sorted(cols))
    indptr = np.cumsum([0] + [len(r) for r in rows_m2]).astype(np.int32)
    indices = np.concatenate(rows_m2).astype(np.int32)
    data = rng.random(indices.shape[0], dtype=np.float32) + 0.5
    m2 = sp.csr_array((data, indices, indptr), shape=(nseg, n_cols))
    m1 = sp.csr_array((np.linspace(1.0, 2.0, nseg).astype(np.float32), np.arange(nseg, dtype=np.int32), np.array([0, nseg], dtype=np.int32)), shape=(1, nseg))
    call = _host.prepare(m1, m2, k=5000)
    r = _host.run_hip(call, time_kernel=True)
    n = r[3][0]
    w = so.run_kernel(call, "port")
    wn = int((w[2] != 0).sum())
    gc = r[1][:n]
    print("shared", shared_cols, "hip entries", n, "distinct", len(set(gc.tolist())), "port entries", wn, "n_dup", r[4]["phase_cycles"][2], "n_q", r[4]["passes_total"])
run([5])
run([40000])
run([5, 40000, 25000])
run([5], seglen=100)
run([5], seglen=100, nseg=8)
run([5], seglen=30, nseg=2)
