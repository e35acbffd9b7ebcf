import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
def _rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32, random_state=np.random.default_rng(seed))
m = _rand((30000, 2000), 0.004, 8)
t = 231
call = _host.prepare(m, k=2000, target_rows=[t])
r = _host.run_hip(call, time_kernel=True)
n = r[3][0]
gc, gv = r[1][:n], r[2][:n]
w = so.run_kernel(call, "port")
wn = int((w[0] == t).sum()) if t else None
wc, wv = w[1][:wn], w[2][:wn]
print("row nnz", call.m1_indptr[t+1]-call.m1_indptr[t], "hip count", n, "port count", wn, "distinct hip cols", len(set(gc.tolist())))
import collections
cnt = collections.Counter(gc.tolist())
multi = {c: sorted(gv[gc == c].tolist()) for c, k_ in cnt.items() if k_ > 1}
wd = dict(zip(wc.tolist(), wv.tolist()))
for c, vs in list(multi.items())[:8]:
    print("col", c, "hip entries", vs, "sum", sum(vs), "port", wd.get(c))
print("info", r[4])
