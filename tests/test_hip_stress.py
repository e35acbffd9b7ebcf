"""GPU tier: the race class under the driver's eyes.

Round 3 removed workgroup barriers from the sparse row kernel and two LDS races came with it: one wrong row
in one of ~7 runs of ONE fuzz case, invisible to every deterministic parity test.  Two things catch that class:

  * the SAME call repeated many times against ONE oracle result (an intermittent difference is a race in the kernels' LDS
    protocol, not arithmetic) — the shapes of scripts/stress_repeat.py: monotone and general variant of the sparse row
    kernel, the 256- and the 1024-thread workgroup shape, the candidate buffer U in LDS and in global memory (the two
    configurations of commit 5b758b7), with and without a MATRIX filter;
  * fixed seeds of the randomised sweep (scripts/fuzz_parity.py as a library): seed 45 (whose case 158 showed the race),
    seed 34, and one seed no earlier round has run.

The oracle (oracle/) is the checker (s_plus.h:39-64 TopK, :192-208 the candidate walk whose semantics the selections keep).
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))

pytestmark = pytest.mark.gpu

REPS = int(os.environ.get("SIMILARIPY_AMD_STRESS_REPS", "100"))
PAD_COL = np.iinfo(np.int32).max


def _rand(shape, density, seed):
    return sp.random_array(shape, density=density, format="csr", dtype=np.float32, random_state=np.random.default_rng(seed))


def _sorted_slots(cols, vals, counts, n, k):
    """(cols[n, k], vals[n, k]) with every slot sorted by column and the padding pushed behind it."""
    c = cols.reshape(n, k).copy()
    v = vals.reshape(n, k).copy()
    pad = np.arange(k)[None, :] >= counts[:, None]
    c[pad] = PAD_COL
    o = np.argsort(c, axis=1, kind="stable")
    return np.take_along_axis(c, o, 1), np.take_along_axis(v, o, 1)


def _slot_lists(c, v, slots):
    return [(c[s][c[s] != PAD_COL], v[s][c[s] != PAD_COL]) for s in slots]


_M = {}


def _matrix(name):
    if name not in _M:
        if name == "wide":
            _M[name] = _rand((55193, 3850), 0.005, 1)
        elif name == "mid":
            _M[name] = _rand((40000, 2000), 0.004, 3)
        elif name == "duo":      # 200 k output columns, ~20 k products per row: the two-per-CU shape of the sparse kernel (round 6)
            from similaripy_amd.workloads import fixed_degree_csr
            _M[name] = fixed_degree_csr(200_000, 10_000, 32, 31)
        else:
            _M[name] = sp.random_array((40000, 40000), density=30.0 / 40000, format="csr", dtype=np.float32, random_state=np.random.default_rng(4))
    return _M[name]


CASES = [
    # name, matrix, kernel kwargs, tunings ({} = the library's choice: 1024 threads here; table_slots=4096 at 1024 threads puts U in global memory)
    ("general splus k=200 binary", "wide", dict(k=200, l1=0.28, l2=0.07, t1=0.51, t2=0.86, c1=0.04, c2=0.29, bayesian_shrink=0.5, a1=2.0, binary=True),
     ({}, {"threads_per_wg": 256}, {"table_slots": 4096})),
    ("monotone cosine k=100", "wide", dict(k=100, l2=1.0, c1=0.5, c2=0.5), ({}, {"threads_per_wg": 256}, {"table_slots": 4096})),
    ("monotone dot k=1000", "wide", dict(k=1000), ({}, {"threads_per_wg": 256})),
    ("monotone cosine + MATRIX filter", "mid", dict(k=50, l2=1.0, c1=0.5, c2=0.5, filter_cols="filter"), ({}, {"threads_per_wg": 256})),
    ("general tversky + shrink", "mid", dict(k=100, l1=1.0, t1=0.4, t2=0.7, stabilized_shrink=3.0), ({}, {"threads_per_wg": 256}, {"table_slots": 4096})),
    # round 6: two 512-thread workgroups per CU (the race of its first build — waves sizing a stage from counters a sibling already pushed to —
    # showed once in 10^6 rows; 100 repeats x 3 000 rows here), monotone and bounded variant; dbg 524288: the classic shape on the same rows
    ("duo monotone cosine k=100", "duo", dict(k=100, l2=1.0, c1=0.5, c2=0.5), ({}, {"dbg": 524288})),
    ("duo bounded splus k=60", "duo", dict(k=60, l1=0.5, l2=0.5, stabilized_shrink=10.0), ({},)),
]


@pytest.mark.parametrize("name,mat,kw,tunings", CASES, ids=[c[0].replace(" ", "_") for c in CASES])
def test_repeated_calls_agree_with_one_oracle_result(name, mat, kw, tunings):
    import torch
    from oracle import splus_oracle as so
    from similaripy_amd import _host
    from similaripy_amd.device import DeviceProblem

    m = _matrix(mat)
    kw = dict(kw)
    if kw.get("filter_cols") == "filter":
        kw["filter_cols"] = _matrix("filter")
    rng = np.random.default_rng(5)
    tg = np.sort(rng.choice(m.shape[0], size=3000, replace=False)).astype(np.int32)
    call = _host.prepare(m, None, target_rows=tg, **kw)
    n, k = call.n_targets, call.k
    wr, wc, wv = so.run_kernel(call, "port")
    wcnt = so.slot_counts(wr, wc, wv, call.targets, k)[0]
    WC, WV = _sorted_slots(wc, wv, wcnt, n, k)

    prob = DeviceProblem(call)                       # operands resident: the repeats time the kernels, not PCIe
    cols, vals, counts, _ = prob.alloc_outputs()
    dev = cols.device
    WC_t, WV_t, wcnt_t = torch.from_numpy(WC).to(dev), torch.from_numpy(WV).to(dev), torch.from_numpy(wcnt).to(dev)
    slot_pos = torch.arange(k, device=dev, dtype=torch.int32)[None, :]
    sparse_rows = 0
    for tun in tunings:
        info = prob.run(cols, vals, counts, time_kernel=True, **tun)
        sparse_rows += info["phase_cycles"][9]
        for rep in range(REPS):
            prob.run(cols, vals, counts, **tun)
            # compared where the result is (sorted slot by slot on the device): a repeat costs the kernels plus a few launches
            c2 = torch.where(slot_pos >= counts[:, None], torch.full_like(cols.view(n, k), PAD_COL), cols.view(n, k))
            GC_t, order = torch.sort(c2, dim=1, stable=True)
            GV_t = torch.gather(vals.view(n, k), 1, order.to(torch.int64))
            same = (GC_t == WC_t).all(dim=1) & torch.isclose(GV_t, WV_t, rtol=2e-5, atol=1e-7).all(dim=1) & (counts == wcnt_t)
            if bool(same.all().item()):
                continue
            if rep > 0:
                # heavily tied data (binary): the k-th place resolves differently from run to run.  A slot whose kept VALUES (sorted)
                # are the oracle's has lost nothing — a race shows as a missing candidate, i.e. a smaller value in its place; the
                # first repeat of every tuning went through the full tie-aware comparator below
                sv_g = torch.sort(torch.where(slot_pos >= counts[:, None], torch.zeros_like(GV_t), vals.view(n, k)), dim=1).values
                sv_w = torch.sort(torch.where(slot_pos >= wcnt_t[:, None], torch.zeros_like(WV_t), WV_t), dim=1).values
                same = same | (torch.isclose(sv_g, sv_w, rtol=2e-5, atol=1e-7).all(dim=1) & (counts == wcnt_t))
                if bool(same.all().item()):
                    continue
            odd = np.flatnonzero(~same.cpu().numpy())    # a k-th place tie may resolve differently: the tie-aware comparator decides
            GC, GV = GC_t.cpu().numpy(), GV_t.cpu().numpy()
            so.compare_topk(_slot_lists(GC, GV, odd), _slot_lists(WC, WV, odd), k, rtol=2e-5, atol=1e-7,
                            what=f"{name} {tun} repeat {rep} (slots {odd[:8].tolist()})")
    assert sparse_rows > 0, "the sparse row kernel did not run: the stress case no longer covers what it is for"


@pytest.mark.parametrize("seed", [45, 34, 404])
def test_fuzz_seed(seed):
    """150 cases of the randomised sweep per seed, the oracle as checker (scripts/fuzz_parity.py)."""
    import fuzz_parity

    stats, failures = fuzz_parity.run_seed(seed, 150, verbose=False)
    assert not failures, "\n".join(failures)
    assert stats["ok"] >= 130, stats
