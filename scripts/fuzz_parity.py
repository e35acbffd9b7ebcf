"""Randomised parity sweep on the GPU: HIP kernels (through the C ABI) against the oracle port on seeded random cases —
shapes that hit both row kernels, every epilogue family, shrinks, thresholds, signed values, ties (binary / quantised
data), target rows, ARRAY / MATRIX selectors, explicit and implicit m2, small tiles (forces windows / give-ups).
Tie-aware comparison as in tests/test_hip_parity.py (identical sets where untied, values within 1e-5 relative).
    python scripts/fuzz_parity.py --cases 300 --seed 1
As a library (tests/test_hip_stress.py):  run_seed(seed, cases) -> (stats, failures); the case sequence of a seed is the
same either way.  Test infrastructure: the oracle is only the checker here."""
import argparse, sys, time, traceback, types
from pathlib import Path
import numpy as np, scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import splus_oracle as so
from similaripy_amd import _abi, _host



def _parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run only this case of the seed's sequence (the others are generated and skipped)")
    ap.add_argument("--dump-slot", type=int, default=-1, help="with --only: print both sides of this slot")
    ap.add_argument("--tuning", default="", help="with --only: override the case's tuning, e.g. table_slots=2048,no_sparse_path=1")
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--huge", action="store_true", help="add shapes with more than 2^18 output columns (changes the case sequence of a seed)")
    ap.add_argument("--duo", action="store_true", help="add shapes whose rows have the headline's weight — 15 k to 40 k products over 4e5 .. 1.5e6 columns: the two-per-CU shape of the sparse kernel (changes the case sequence of a seed)")
    ap.add_argument("--max-macs", type=float, default=4e8, help="skip cases whose oracle run would take too long")
    return ap


# the options and the generator of the running sweep (set by run_seed)
a = _parser().parse_args([])
rng = np.random.default_rng(a.seed)


def _run_dbg(call, tuning, dbg):
    from similaripy_amd.device import DeviceProblem
    prob = DeviceProblem(call)
    c, v, n, r = prob.alloc_outputs(with_rows=True)
    prob.run(c, v, n, rows=r, dbg=dbg, **tuning)
    import torch; torch.cuda.synchronize()
    return r.cpu().numpy(), c.cpu().numpy(), v.cpu().numpy(), n.cpu().numpy()


def rand_matrix(n_rows, n_cols, density, kind):
    m = sp.random_array((n_rows, n_cols), density=density, format="csr", dtype=np.float32, random_state=rng)
    if kind == "binary":
        m.data[:] = 1.0
    elif kind == "quant":
        m.data[:] = np.round(m.data * 4 + 1) / 4           # heavy ties
    elif kind == "signed":
        m.data[:] = (m.data - 0.5) * 2
    elif kind == "skewed":                                   # popular columns / long rows
        cols = (rng.pareto(1.2, size=m.nnz) * n_cols / 50).astype(np.int64) % n_cols
        m = sp.csr_array((m.data, cols.astype(np.int32), m.indptr), shape=m.shape)
        m.sum_duplicates()
    return m


def one_case(i):
    kinds = ["small", "wide_out", "tall", "dense_rows"] + (["huge_out"] if a.huge else []) + (["duo_out", "duo_out"] if getattr(a, "duo", False) else [])
    shape_kind = rng.choice(kinds)
    if shape_kind == "small":
        n_rows, n_cols, dens = int(rng.integers(1, 400)), int(rng.integers(1, 300)), float(rng.choice([0.02, 0.1, 0.4]))
    elif shape_kind == "wide_out":          # m2 = m.T has many columns: the sparse kernel
        n_rows, n_cols, dens = int(rng.integers(20000, 60000)), int(rng.integers(500, 4000)), float(rng.choice([0.002, 0.005, 0.01]))
    elif shape_kind == "huge_out":          # explicit m2 with more than 2^18 columns: the aliasing bitmap of the small shape
        n_rows, n_cols, dens = int(rng.integers(3000, 8000)), int(rng.integers(300, 800)), float(rng.choice([0.01, 0.03]))
    elif shape_kind == "duo_out":           # explicit m2 with 4e5 .. 1.5e6 columns and rows of 300 .. 700 entries: 15 k .. 40 k products per target row
        n_rows, n_cols = int(rng.integers(1500, 4000)), int(rng.integers(2000, 4000))
        dens = float(rng.integers(40, 65)) / n_cols
    elif shape_kind == "tall":
        n_rows, n_cols, dens = int(rng.integers(3000, 9000)), int(rng.integers(50, 400)), float(rng.choice([0.02, 0.08]))
    else:
        n_rows, n_cols, dens = int(rng.integers(200, 1500)), int(rng.integers(2000, 8000)), float(rng.choice([0.02, 0.05]))
    kind = str(rng.choice(["plain", "plain", "binary", "quant", "signed", "skewed"]))
    m = rand_matrix(n_rows, n_cols, dens, kind)
    explicit_m2 = rng.random() < 0.3 or shape_kind in ("huge_out", "duo_out")
    m2 = None
    if explicit_m2:
        nc2 = int(rng.integers(1, 5000)) if shape_kind != "wide_out" else int(rng.integers(20000, 50000))
        if shape_kind == "huge_out":
            nc2 = int(rng.integers(300_000, 900_000))
        if shape_kind == "duo_out":
            nc2 = int(rng.integers(400_000, 1_500_000))
            m2 = rand_matrix(n_cols, nc2, float(rng.integers(300, 700)) / nc2, str(rng.choice(["plain", "plain", "signed", "quant"])))
        else:
          m2 = rand_matrix(n_cols, nc2, (float(rng.choice([0.00005, 0.0002, 0.0005])) if nc2 > 100_000 else float(rng.choice([0.002, 0.01, 0.05]))) if nc2 > 1000 else 0.1, str(rng.choice(["plain", "signed", "quant"])))
    n_out = m.shape[0] if m2 is None else m2.shape[1]
    fam = str(rng.choice(["dot", "cosine", "asym", "tversky", "jaccard", "dice", "splus", "depop", "rp3like"]))
    kw = {}
    if fam == "cosine": kw.update(l2=1.0)
    elif fam == "asym": kw.update(l2=1.0, c1=float(rng.random()), c2=float(rng.random()))
    elif fam == "tversky": kw.update(l1=1.0, t1=float(rng.random()), t2=float(rng.random()))
    elif fam == "jaccard": kw.update(l1=1.0, t1=1.0, t2=1.0)
    elif fam == "dice": kw.update(l1=1.0, t1=0.5, t2=0.5)
    elif fam == "splus": kw.update(l1=float(rng.random()), l2=float(rng.random()), t1=float(rng.random()), t2=float(rng.random()), c1=float(rng.random()), c2=float(rng.random()))
    elif fam == "depop": kw.update(l1=0.3, l2=0.3, l3=0.4, weight_depop_matrix1="sum", weight_depop_matrix2="sum", p1=float(rng.random()), p2=float(rng.random()))
    elif fam == "rp3like": kw.update(l3=1.0, weight_depop_matrix2="sum", p2=float(rng.random()))
    if fam != "dot" and rng.random() < 0.4:
        kw[str(rng.choice(["stabilized_shrink", "bayesian_shrink", "additive_shrink"]))] = float(rng.choice([0.5, 3.0, 20.0]))
    thr_draw, thr_val = rng.random(), float(rng.choice([0.0, 1e-3, 0.05, 0.3, -0.1]))
    if thr_draw < 0.25 and kind not in ("quant", "binary") and not kw.get("binary"):     # (quantised values sit ON round thresholds: which side a value falls is rounding)
        kw["threshold"] = thr_val
    if rng.random() < 0.15: kw["a1"] = float(rng.choice([0.5, 2.0]))
    if rng.random() < 0.2:
        kw["binary"] = True
        kw.pop("threshold", None)          # (binary data: values sit on round thresholds, see above)
    k = int(rng.choice([1, 5, 10, 50, 100, 200, 1000]))
    n_t = int(min(m.shape[0], rng.choice([m.shape[0], 50, 300, 1500])))
    if shape_kind == "duo_out":
        n_t = min(n_t, 600)                 # (bounds the oracle: 600 rows x 40 k products)
    targets = None if n_t == m.shape[0] and rng.random() < 0.5 else np.sort(rng.choice(m.shape[0], size=n_t, replace=False)).astype(np.int32)
    sel = rng.random()
    if shape_kind in ("huge_out", "duo_out") and sel >= 0.2:
        sel = 1.0            # (no MATRIX selectors over ~1e6 columns x ~1e4 rows here)
    if sel < 0.12: kw["filter_cols"] = rng.choice(n_out, size=max(1, n_out // 7), replace=False).tolist()
    elif sel < 0.2: kw["target_cols"] = rng.choice(n_out, size=max(1, n_out // 3), replace=False).tolist()
    elif sel < 0.32: kw["filter_cols"] = sp.random_array((m.shape[0], n_out), density=min(0.5, 30.0 / max(n_out, 1)), format="csr", dtype=np.float32, random_state=rng)
    elif sel < 0.4: kw["target_cols"] = sp.random_array((m.shape[0], n_out), density=min(0.9, 200.0 / max(n_out, 1)), format="csr", dtype=np.float32, random_state=rng)
    tuning = {}
    if rng.random() < 0.2: tuning["table_slots"] = int(rng.choice([1024, 2048, 4096]))
    if rng.random() < 0.15: tuning["threads_per_wg"] = int(rng.choice([256, 512, 768]))
    # (a generator of its own, as for `stages` below: one case in five asks for the wave-per-row kernel, which the library grants wherever the
    # call qualifies — monotone epilogue, k <= 128 — and whose rows of more than 64 entries or 10 k products take the other queues)
    if not tuning and np.random.default_rng([a.seed, i, 11]).random() < 0.2: tuning["threads_per_wg"] = 64
    on_dev = (m2 is None) and rng.random() < 0.6
    # (a generator of its own: the cases of the seeds of earlier rounds stay what they were)
    stages = bool(np.random.default_rng([a.seed, i, 7]).random() < 0.4) and not a.dbg
    desc = f"#{i} {shape_kind} {m.shape} kind={kind} m2={'explicit ' + str(m2.shape) if m2 is not None else 'm1.T' + ('(device)' if on_dev else '')} fam={fam} k={k} targets={'all' if targets is None else len(targets)}{' host-mode stages on the device' if stages else ''} kw={ {x: (v if not hasattr(v, 'shape') and not isinstance(v, list) else type(v).__name__) for x, v in kw.items()} } tuning={tuning}"
    if a.only >= 0 and i != a.only:
        return "skipped", desc
    if a.tuning:
        tuning = {x.split("=")[0]: int(x.split("=")[1]) for x in a.tuning.split(",")}
    call = _host.prepare(m, m2, k=k, target_rows=targets, m2_on_device=on_dev, **kw)
    # bound the oracle's work
    ref_call = call
    check_zeros = False
    if stages:
        # what the public calls do: norms, the ones of binary=True, ARRAY selectors, the zero count and the look at the order inside the
        # rows of an explicit m2 are left to the library's host-mode entry (SP_FLAG_NORMS_ON_DEVICE / BINARY / CHECK_ZEROS / CHECK_SORTED,
        # col_keep); the oracle gets the host statement of the same stages
        call = _host.prepare(m, m2, k=k, target_rows=targets, m2_on_device=True, norms_on_device=True, binary_on_device=True, m2_sorted_on_device=True,
                             keep_on_device=True, check_zeros=False, **kw)
        ref_call = _host.prepare(m, m2, k=k, target_rows=targets, m2_on_device=False, **kw)
        check_zeros = True
    elif call.m2_is_m1t:
        import dataclasses
        t = sp.csr_array((call.m1_data, call.m1_indices, call.m1_indptr), shape=(call.n_rows_m1, call.n_rows_m2)).T.tocsr(); t.sort_indices()
        d2, i2, p2 = np.ascontiguousarray(t.data, np.float32), np.ascontiguousarray(t.indices, np.int32), np.ascontiguousarray(t.indptr, np.int32)
        if call.col_keep is not None:      # ARRAY selectors: the library drops those columns while it builds m2 (sp_knn_args.col_keep)
            d2, i2, p2 = _host.filter_matrix_columns(d2, i2, p2, call.n_output_cols, np.flatnonzero(call.col_keep).astype(np.int32))
        ref_call = dataclasses.replace(call, m2_data=d2, m2_indices=i2, m2_indptr=p2, m2_is_m1t=False, col_keep=None)
    colnnz = np.diff(ref_call.m2_indptr)
    macs = float(colnnz[ref_call.m1_indices].sum()) * (len(ref_call.targets) / max(1, ref_call.n_rows_m1))
    if macs > a.max_macs:
        return "skipped", desc
    try:
        rows, cols, vals, counts = _host.run_hip(call, check_zeros=check_zeros, **tuning) if not a.dbg else _run_dbg(call, tuning, a.dbg)
    except (_abi.ExplicitZerosError, _abi.UnsortedRowsError):
        # (what the public call does next: the host removes the zeros / sorts a copy and calls again)
        call = _host.prepare(m, m2, k=k, target_rows=targets, m2_on_device=True, norms_on_device=True, binary_on_device=True, keep_on_device=True, **kw)
        rows, cols, vals, counts = _host.run_hip(call, **tuning)
    got = so.canonical(rows, cols, vals, call.targets, call.k)
    want = so.canonical(*so.run_kernel(ref_call, "port"), call.targets, call.k)
    # Reference quirk (s_plus.h:112-116): `add()` takes "running sum == 0" for "first touch", so a column whose partial
    # sum is exactly 0 when its next product arrives is listed twice and emitted a second time with the value of xy = 0.
    # The HIP kernels emit every column once, with its full sum: drop the reference's zero-valued duplicate.
    for s_, (wc_, wv_) in enumerate(want):
        if wc_.shape[0] > 1 and (np.diff(wc_) == 0).any():
            keep = np.ones(wc_.shape[0], dtype=bool)
            for c_ in np.unique(wc_[:-1][np.diff(wc_) == 0]):
                idx = np.flatnonzero(wc_ == c_)
                best = idx[np.argmax(np.abs(wv_[idx]))]
                keep[idx] = False
                keep[best] = True
            want[s_] = (wc_[keep], wv_[keep])
    # The same quirk with the FIRST emission dropped by the threshold (a negative value against threshold >= 0): only the duplicate
    # with value exactly 0 is left in the reference's result.  Told apart from a genuine zero (a raw dot that cancels exactly, a zero
    # denominator) by the column's raw dot in float64: non-zero -> the reference's artefact, dropped here.
    if not (kw.get("binary") or kind == "binary"):
        M1 = sp.csr_array((ref_call.m1_data, ref_call.m1_indices, ref_call.m1_indptr), shape=(ref_call.n_rows_m1, ref_call.n_rows_m2)).astype(np.float64)
        M2 = None
        for s_, ((gc_, gv_), (wc_, wv_)) in enumerate(zip(got, want)):
            odd = np.flatnonzero((wv_ == 0) & ~np.isin(wc_, gc_))
            if odd.shape[0] == 0:
                continue
            if M2 is None:
                M2 = sp.csr_array((ref_call.m2_data, ref_call.m2_indices, ref_call.m2_indptr), shape=(ref_call.n_rows_m2, ref_call.n_output_cols)).astype(np.float64).tocsc()
            xy = np.asarray((M1[[int(call.targets[s_])]] @ M2[:, wc_[odd]]).todense()).ravel()
            terms_ok = np.ones(odd.shape[0], dtype=bool)
            for y_ in (ref_call.Ycosine, ref_call.Ydepop):
                if y_ is not None and y_.shape[0] > 0:
                    terms_ok &= (y_[wc_[odd]] != 0)
            drop = odd[(xy != 0) & terms_ok]
            if drop.shape[0]:
                keep = np.ones(wc_.shape[0], dtype=bool); keep[drop] = False
                want[s_] = (wc_[keep], wv_[keep])
    if a.dump_slot >= 0:
        gc, gv = got[a.dump_slot]; wc, wv = want[a.dump_slot]
        print(desc); print("got ", dict(zip(gc.tolist(), gv.tolist()))); print("want", dict(zip(wc.tolist(), wv.tolist())))
        miss = sorted(set(wc.tolist()) - set(gc.tolist())); print("missing", miss, "target row", call.targets[a.dump_slot])
        t = int(call.targets[a.dump_slot])
        us = call.m1_indices[call.m1_indptr[t]:call.m1_indptr[t + 1]]
        allc = np.concatenate([ref_call.m2_indices[ref_call.m2_indptr[u]:ref_call.m2_indptr[u + 1]] for u in us]) if len(us) else np.zeros(0, np.int32)
        uq, cn = np.unique(allc, return_counts=True)
        print("row has", len(us), "m1 entries,", allc.shape[0], "products,", uq.shape[0], "distinct columns; products per missing column:", {int(c): int(cn[uq == c][0]) for c in miss},
              "; multi-product columns:", int((cn > 1).sum()), "segment lengths", [int(ref_call.m2_indptr[u + 1] - ref_call.m2_indptr[u]) for u in us])
        import dataclasses
        one = dataclasses.replace(call, targets=np.array([t], dtype=np.int32))
        r1, c1, v1, n1 = _host.run_hip(one, **tuning)
        print("alone: kept", int(n1[0]), "missing", sorted(set(wc.tolist()) - set(c1[:n1[0]].tolist())))
        info = _host.run_hip(one, time_kernel=True, **tuning)[4]
        print("alone: rows sparse / given up:", info["phase_cycles"][9], info["phase_cycles"][10])
    # signed data: sums cancel, and a different (equally valid) summation order moves a value by more than 1e-5 of itself
    signed = kind == "signed" or (m2 is not None and bool((m2.data < 0).any()))
    # ... and with a Bayesian shrink b the value has a pole at raw dot = -b: near it no tolerance is meaningful
    pole = signed and (kw.get("bayesian_shrink", 0.0) != 0.0 or kw.get("l1", 0.0) != 0.0)      # (a Tversky denominator has one too)
    pole_ = pole      # (values there may differ by any factor: 1e9 = sets only)
    if pole:
        # ... and a row that sits ON the pole on either side (a denominator of exactly 0 in one summation order: inf, or a value beyond any the data can
        # produce away from it) has no defined top-k at all: set aside (seed 408 case 161: raw dot = -0.5 = -b exactly on one side, +inf kept first)
        empty = (np.zeros(0, np.int32), np.zeros(0, np.float32))
        for j in range(len(got)):
            gv_, wv_ = got[j][1], want[j][1]
            if (gv_.size and (~np.isfinite(gv_) | (np.abs(gv_) > 1e6)).any()) or (wv_.size and (~np.isfinite(wv_) | (np.abs(wv_) > 1e6)).any()):
                got[j] = want[j] = empty
    so.compare_topk(got, want, call.k, rtol=(1e9 if pole else 1e-3) if signed else 1e-5, atol=1e-5 if signed else 1e-7, what=desc, threshold=kw.get("threshold"))
    n, kk = call.n_targets, call.k
    pad = np.arange(kk)[None, :] >= counts[:, None]
    assert not rows.reshape(n, kk)[pad].any() and not cols.reshape(n, kk)[pad].any() and not vals.reshape(n, kk)[pad].any(), "padding not zero: " + desc
    return "ok", desc



def run_seed(seed, cases, verbose=True, **opts):
    """The first `cases` cases of `seed`'s sequence.  Returns (stats, [messages of the failed cases])."""
    global a, rng
    a = _parser().parse_args([])
    a.seed, a.cases = int(seed), int(cases)
    for k_, v_ in opts.items():
        setattr(a, k_, v_)
    rng = np.random.default_rng(a.seed)
    stats = {"ok": 0, "skipped": 0, "failed": 0}
    failures = []
    for i in range(a.cases):
        try:
            r, desc = one_case(i)
            stats[r] += 1
        except Exception as exc:      # keep going: report every failing case with what it takes to reproduce it
            stats["failed"] += 1
            failures.append(f"case {i} (seed {a.seed}): {type(exc).__name__}: {str(exc)[:600]}")
            if verbose:
                print("FAILED " + failures[-1], flush=True)
                if not isinstance(exc, AssertionError):
                    traceback.print_exc()
    return stats, failures


def main():
    o = _parser().parse_args()
    t0 = time.time()
    stats, _ = run_seed(o.seed, o.cases, only=o.only, dump_slot=o.dump_slot, tuning=o.tuning, dbg=o.dbg, huge=o.huge, duo=o.duo, max_macs=o.max_macs)
    print(f"fuzz: {stats} in {time.time() - t0:.0f}s (seed {o.seed})")
    sys.exit(1 if stats["failed"] else 0)


if __name__ == "__main__":
    main()
