import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, scipy.sparse as sp
from similaripy_amd import _host
from oracle import splus_oracle as so
def _rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32, random_state=np.random.default_rng(seed))
for shape, density, k in (((40000, 2000), 0.005, 50), ((1500, 2500), 0.04, 30), ((3000,3000),0.01,10)):
    rng = np.random.default_rng(7)
    m = _rand(shape, density, 7).tolil()
    zero_rows = rng.choice(shape[0], size=shape[0] // 10, replace=False)
    for j in zero_rows:
        cols = rng.choice(shape[1], size=4, replace=False)
        m[j, :] = 0
        m[j, cols[0]] = 0.5; m[j, cols[1]] = -0.5; m[j, cols[2]] = 0.25; m[j, cols[3]] = -0.25
    m = m.tocsr().astype(np.float32); m.eliminate_zeros(); m.sort_indices()
    targets = np.sort(rng.choice(shape[0], size=min(shape[0], 400), replace=False)).astype(np.int32)
    for thr in (0.0, -1.0):
        call = _host.prepare(m, k=k, l3=1.0, weight_depop_matrix2="sum", p2=0.5, target_rows=targets, threshold=thr)
        want = so.canonical(*so.run_kernel(call, "port"), call.targets, call.k)
        for dbg in (0, 2097152):
            rows, cols, vals, counts = _host.run_hip(call, dbg=dbg)
            got = so.canonical(rows, cols, vals, call.targets, call.k)
            try:
                t = so.compare_topk(got, want, call.k, rtol=1e-5, atol=1e-7, what=f"{shape} thr={thr} dbg={dbg}")
                print(shape, thr, dbg, "OK ties", t)
            except AssertionError as e:
                print(shape, thr, dbg, "FAIL", str(e)[:300])
