import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
def _rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32, random_state=np.random.default_rng(seed))
m = _rand((30000, 2000), 0.004, 8)
call = _host.prepare(m, k=50, target_rows=np.arange(0, 30000, 7))
want = so.canonical(*so.run_kernel(call, "port"), call.targets, 50)
for kw in (dict(no_sparse_path=True), dict(threads_per_wg=1024), dict(threads_per_wg=512), dict(threads_per_wg=256), dict(table_slots=16384)):
    r = _host.run_hip(call, table_slots=kw.pop("table_slots", 4096), **kw)
    got = so.canonical(r[0], r[1], r[2], call.targets, 50)
    bad = sum(1 for (gc, gv), (wc, wv) in zip(got, want) if gc.shape != wc.shape or not np.array_equal(gc, wc) or not np.allclose(gv, wv, rtol=1e-5))
    print(kw, "bad slots:", bad)
