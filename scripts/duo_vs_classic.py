"""Where does the two-per-CU shape of the sparse kernel (round 6) win?  A grid of fixed-degree shapes around configs[1] — columns of m1, entries per
row, k, epilogue — each timed with the library's own choice and with the shape switched off (ablation bit 524288), 200 k target rows, kernel scope.
`python scripts/duo_vs_classic.py > gpurun_out/r06_exp_duo_vs_classic.txt`"""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
from similaripy_amd import _host
from similaripy_amd.device import DeviceProblem
from similaripy_amd.workloads import fixed_degree_csr

def timed(prob, outs, **tun):
    prob.run(*outs, **tun); torch.cuda.synchronize()
    ms = [prob.run(*outs, time_kernel=True, phase_timers=False, **tun) for _ in range(3)]
    i = prob.run(*outs, time_kernel=True, **tun)
    return min(m["sparse_kernel_ms"] + m["generic_kernel_ms"] for m in ms), i["num_wgs"], i["phase_cycles"][10] & 0xFFFFFFFF

print(f"{'rows x cols(m1), nnz/row':28s} {'epilogue':10s} {'k':>5s} {'MACs/row':>9s}  {'chosen ms':>9s} {'wgs':>4s} {'fallb':>5s}   {'classic ms':>10s} {'fallb':>5s}   ratio")
grid = [  # (n_rows, n_cols_m1, nnz_row)
    (1_000_000, 100_000, 64), (1_000_000, 100_000, 48), (1_000_000, 100_000, 32), (1_000_000, 200_000, 64), (1_000_000, 50_000, 32),
    (400_000, 40_000, 64), (2_000_000, 200_000, 64), (4_000_000, 400_000, 64), (300_000, 30_000, 48),
]
for n_rows, n_cols, nnz in grid:
    m = fixed_degree_csr(n_rows, n_cols, nnz, 7)
    t = np.arange(0, min(n_rows, 200_000), dtype=np.int32)
    for name, kw, k in (("cosine", dict(l2=1, c1=.5, c2=.5), 100), ("cosine", dict(l2=1, c1=.5, c2=.5), 10), ("cosine", dict(l2=1, c1=.5, c2=.5), 1000),
                        ("splus", dict(l1=.5, l2=.5, stabilized_shrink=10.0), 100)):
        if (n_rows, nnz) != (1_000_000, 64) and (k != 100):
            continue
        call = _host.prepare(m, k=k, target_rows=t, **kw)
        prob = DeviceProblem(call)
        outs = prob.alloc_outputs()[:3]
        a = timed(prob, outs)
        b = timed(prob, outs, dbg=524288)
        macs = nnz * nnz * n_rows / n_cols
        print(f"{f'{n_rows} x {n_cols}, {nnz}':28s} {name:10s} {k:5d} {macs:9.0f}  {a[0]:9.2f} {a[1]:4d} {a[2]:5d}   {b[0]:10.2f} {b[2]:5d}   {b[0] / a[0]:.2f}", flush=True)
        del prob, outs
    del m
    torch.cuda.empty_cache()
