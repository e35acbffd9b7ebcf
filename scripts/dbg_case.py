import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from similaripy_amd import _host
from oracle import splus_oracle as so
def _rand(shape, density, seed, dtype=np.float32):
    return sp.random_array(shape, density=density, format="csr", dtype=dtype, random_state=np.random.default_rng(seed))
m = _rand((30000, 2000), 0.004, 8)
for T in (4096, 1024, 0):
    for kw in ({}, dict(l2=1)):
        call = _host.prepare(m, k=50, target_rows=np.arange(0, 30000, 7), **kw)
        rows, cols, vals, counts, info = _host.run_hip(call, table_slots=T, time_kernel=True)
        k = call.k
        got = so.canonical(rows, cols, vals, call.targets, k)
        want = so.canonical(*so.run_kernel(call, "port"), call.targets, k)
        bad = 0
        for i, ((gc, gv), (wc, wv)) in enumerate(zip(got, want)):
            gs, ws = dict(zip(gc.tolist(), gv.tolist())), dict(zip(wc.tolist(), wv.tolist()))
            if set(gs) != set(ws) or any(abs(gs[c]-ws[c]) > 1e-5*abs(ws[c])+1e-7 for c in gs):
                bad += 1
                if bad <= 3:
                    miss = sorted(set(ws) - set(gs)); extra = sorted(set(gs) - set(ws))
                    print(f"T={T} kw={kw} row slot {i} t={call.targets[i]}: n_got={len(gs)} n_want={len(ws)} missing={[(c, ws[c]) for c in miss][:5]} extra={[(c, gs[c]) for c in extra][:5]} valdiff={[(c, gs[c], ws[c]) for c in gs if c in ws and abs(gs[c]-ws[c])>1e-5*abs(ws[c])+1e-7][:5]} min_got={min(gs.values()) if gs else None} min_want={min(ws.values()) if ws else None}")
        pc = info["phase_cycles"]
        print(f"T={T} kw={kw}: bad rows {bad}/{len(got)}; sparse rows {pc[9]} fallback {pc[10]} windows {pc[11]}")
