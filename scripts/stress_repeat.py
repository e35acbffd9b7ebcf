"""Race hunt: the same call many times against ONE oracle result (tie-aware comparison) — an intermittent difference is a race in the
kernels' LDS protocol, not arithmetic.  Shapes that run the sparse row kernel in its monotone and general variants, the 256- and the
1024-thread shape, with and without a MATRIX filter.   python scripts/stress_repeat.py [repeats]"""
import sys, time
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import splus_oracle as so
from similaripy_amd import _host

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(5)
def rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32, random_state=np.random.default_rng(seed))
cases = []
m = rand((55193, 3850), 0.005, 1)
cases.append(("general splus k=200 binary", m, None, dict(k=200, l1=0.28, l2=0.07, t1=0.51, t2=0.86, c1=0.04, c2=0.29, bayesian_shrink=0.5, a1=2.0, binary=True)))
cases.append(("monotone cosine k=100", m, None, dict(k=100, l2=1.0, c1=0.5, c2=0.5)))
cases.append(("monotone dot k=1000", m, None, dict(k=1000)))
m3 = rand((40000, 2000), 0.004, 3)
f = sp.random_array((40000, 40000), density=30.0 / 40000, format="csr", dtype=np.float32, random_state=np.random.default_rng(4))
cases.append(("monotone cosine + MATRIX filter", m3, None, dict(k=50, l2=1.0, c1=0.5, c2=0.5, filter_cols=f)))
cases.append(("general tversky + shrink", m3, None, dict(k=100, l1=1.0, t1=0.4, t2=0.7, stabilized_shrink=3.0)))
bad = 0
for name, a, b, kw in cases:
    tg = np.sort(rng.choice(a.shape[0], size=6000, replace=False)).astype(np.int32)
    call = _host.prepare(a, b, target_rows=tg, **kw)
    want = so.canonical(*so.run_kernel(call, "port"), call.targets, call.k)
    t0 = time.time()
    for r in range(reps):
        for tun in ({}, {"threads_per_wg": 256}) + (({"threads_per_wg": 64},) if name.startswith("monotone") else ()):      # (64: the wave-per-row kernel where the call qualifies)
            rows, cols, vals, counts = _host.run_hip(call, **tun)
            got = so.canonical(rows, cols, vals, call.targets, call.k)
            try:
                so.compare_topk(got, want, call.k, rtol=2e-5, atol=1e-7, what=name)
            except AssertionError as e:
                bad += 1
                print(f"MISMATCH {name} rep {r} {tun}: {str(e)[:300]}", flush=True)
    print(f"{name}: {reps} x 2 (monotone: x 3) runs in {time.time() - t0:.1f}s", flush=True)
print("stress:", "FAILED" if bad else "ok", bad)
