import numpy as np, scipy.sparse as sp, sys
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
def _rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32, random_state=np.random.default_rng(seed))
m = _rand((30000, 2000), 0.004, 8)
call = _host.prepare(m, k=50, target_rows=np.arange(0, 30000, 7))
rows, cols, vals, counts = _host.run_hip(call, table_slots=4096)
wr, wc, wv = so.run_kernel(call, "port")
k = 50; n = call.n_targets
bad = 0
for i in range(n):
    g = dict(zip(cols[i*k:i*k+counts[i]].tolist(), vals[i*k:i*k+counts[i]].tolist()))
    gl = cols[i*k:i*k+counts[i]]
    wcnt = int((wr[i*k:(i+1)*k] == call.targets[i]).sum())
    w = dict(zip(wc[i*k:i*k+wcnt].tolist(), wv[i*k:i*k+wcnt].tolist()))
    dupcols = len(gl) - len(set(gl.tolist()))
    diffs = [(c, g[c], w[c]) for c in g if c in w and abs(g[c]-w[c]) > 1e-5*abs(w[c])]
    if dupcols or diffs or len(g) != len(w):
        bad += 1
        if bad <= 5:
            print("slot", i, "target", call.targets[i], "count", counts[i], "want", wcnt, "dup columns", dupcols, "diffs", diffs[:6])
print("bad slots", bad, "of", n)
# detail for bad slots
import collections
shown = 0
for i in range(n):
    gl = cols[i*k:i*k+counts[i]]; gv = vals[i*k:i*k+counts[i]]
    cnt = collections.Counter(gl.tolist())
    d = [c for c, m_ in cnt.items() if m_ > 1]
    if d and shown < 3:
        shown += 1
        wcnt = int((wr[i*k:(i+1)*k] == call.targets[i]).sum())
        w = dict(zip(wc[i*k:i*k+wcnt].tolist(), wv[i*k:i*k+wcnt].tolist()))
        t = call.targets[i]
        print("slot", i, "row nnz", call.m1_indptr[t+1]-call.m1_indptr[t])
        for c in d[:6]:
            print("   col", c, "got values", gv[gl == c].tolist(), "oracle", w.get(c))
