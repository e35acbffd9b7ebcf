"""Scratch: a target row whose m1 entries point at EMPTY m2 rows, on the sparse kernel (n_cols > tile)."""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
DBG = False
def run(seglens, T=2048, n_cols=4000, **kw):
    n1 = len(seglens)
    rng = np.random.default_rng(1)
    m1 = sp.csr_array((rng.random(n1, dtype=np.float32) + 0.5, np.arange(n1, dtype=np.int32), np.array([0, n1], dtype=np.int32)), shape=(1, n1))
    indptr = np.concatenate(([0], np.cumsum(seglens))).astype(np.int32)
    cols = np.concatenate([np.sort(rng.choice(n_cols, size=l, replace=False)) for l in seglens] + [np.zeros(0, np.int64)]).astype(np.int32)
    m2 = sp.csr_array((rng.random(cols.shape[0], dtype=np.float32) + 0.1, cols, indptr), shape=(n1, n_cols))
    call = _host.prepare(m1, m2, k=50, **kw)
    rows, c, v, counts, info = _host.run_hip(call, table_slots=T, time_kernel=True)
    if DBG:
        from similaripy_amd import device
        import torch
        prob = device.DeviceProblem(call)
        o = prob.alloc_outputs()
        inf = prob.run(o[0], o[1], o[2], time_kernel=True, table_slots=T, dbg=256)
        pc = inf['phase_cycles']
        print('   lanes (pos,heavy,nit_p,my_ib,len):', [((x>>48)&0xffff, (x>>40)&0xff, (x>>32)&0xff, (x>>16)&0xffff, x&0xffff) for x in pc[:10]], 'H', bin(pc[10]), 'thr', hex(pc[11]>>32), 'ib_incl(lane0)', pc[11]&0xffffffff)
    got = so.canonical(rows, c, v, call.targets, call.k)[0]
    want = so.canonical(*so.run_kernel(call, "port"), call.targets, call.k)[0]
    miss = sorted(set(want[0].tolist()) - set(got[0].tolist()))
    seg_of = {int(cc): s for s in range(n1) for cc in cols[indptr[s]:indptr[s + 1]]}
    print(f"seglens={seglens} kw={kw}: got {len(got[0])} want {len(want[0])} missing from segments {sorted(set(seg_of[x] for x in miss))} sparse/given up {info['phase_cycles'][9]}/{info['phase_cycles'][10]}")
for sl in ([5, 5, 5], [5, 0, 5], [0, 5, 5], [5, 5, 0], [5, 0, 0, 5, 3], [0, 0, 4], [3, 0, 4, 0, 2, 0, 6, 1], [1, 2, 5, 5, 5, 5, 0, 0, 5, 5, 6, 6]):
    run(sl)
run([5, 0, 5], l2=1)
run([5, 0, 5], T=0, n_cols=40000)
