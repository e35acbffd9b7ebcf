"""GPU tier: every BASELINE.json config at its defining shape.

  configs[0]  sim.cosine on sps.random 10k x 20k d=0.01 k=50            every row vs the oracle
  configs[1]  cosine 1M x 100k, 64 nnz/row, k=100                        properties over all rows + 20 000 rows vs the oracle
  configs[2]  s_plus(l1=.5, l2=.5, shrink=10) on the same matrix         20 000 rows vs the oracle, all rows on the sparse kernel
  configs[3]  p3alpha + rp3beta, MovieLens-32M-shaped URM.T, k=200       through the public wrappers; 320 rows (the 20 heaviest
                                                                         included) vs the oracle
  configs[4]  dot_product(urm, W.T, filter_cols=urm), 1M users, k=100    20 000 rows vs the oracle + "nothing seen is recommended"
                                                                         over all rows (one GPU's slice of the 10M-user job)

The oracle cannot finish these sizes in seconds, so full results are checked through size-independent properties and a
sample of rows is compared with the oracle exactly (tie-aware, 1e-5 relative; long float32 sums: see `_check_against_float64`).
Samples of configs[1], [2], [4]: 18 000 random rows + the 2 000 LIGHTEST rows (fewest MACs) — rows are queued heaviest-first
(sp_row_order_kernel), so those are the tail of the work-ordered queue, where the persistent workgroups run dry (VERDICT r4 #5a;
the oracle does ~37 k rows/s on the box's cores)."""
from __future__ import annotations

import copy
import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import similaripy_amd as sim                            # noqa: E402
from oracle import splus_oracle as so                  # noqa: E402
from similaripy_amd import _host, workloads            # noqa: E402
from oracle.norm_oracle import normalize               # noqa: E402  (NumPy statement of the reference's L1 normaliser)

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-7       # north_star: float32 values within 1e-5 relative
N_SAMPLE, N_TAIL = 20_000, 2_000


def _macs_per_row(m1, m2_indptr):
    per = np.diff(m2_indptr).astype(np.int64)[m1.indices]
    csum = np.concatenate(([0], np.cumsum(per)))
    return csum[m1.indptr[1:]] - csum[m1.indptr[:-1]]


def _sample_with_queue_tail(macs, seed):
    """N_SAMPLE distinct rows: the N_TAIL lightest (the end of the work-ordered queue) + random ones."""
    n = macs.shape[0]
    tail = np.argsort(macs, kind="stable")[:N_TAIL]
    rnd = np.random.default_rng(seed).choice(n, N_SAMPLE, replace=False)
    return np.unique(np.concatenate((tail, rnd)))[:N_SAMPLE + N_TAIL].astype(np.int32)


# configs[3]'s bar, stated (DESIGN §5 "configs[3] parity" carries the same text for readers of BASELINE.md):
#   rows of <= 10^6 MACs (94 % of the rows): north_star's own — HIP within 1e-5 relative of the float64 value AND of the reference.
#   heavier rows: their values are float32 sums of 10^5 .. 10^6 products; the REFERENCE's float32 result is itself up to 1.4e-5 from the
#   float64 value there (sqrt(n) * 2^-24 for n = 2e5 is 2.7e-5), so "within 1e-5 of the reference" cannot be asked of two float32 sums in
#   different orders.  What is asked instead, as a fixed regression bound and not a formula over the reference's error: HIP within
#   HEAVY_F64_BOUND = 2e-5 of the FLOAT64 value.  The HIP sums are LDS atomics in whatever order the waves arrive: the same build gives
#   1.27e-5 .. 1.52e-5 on the heaviest rows from run to run (the sequential reference: 1.40e-5 every time; observed over 5 018 rows of each
#   call: profiles/r06_c4_value_errors.txt) — a bound of 1.5e-5, the round's first choice, failed one run in about ten.
#   Column sets: a column on one side only must lie within 2.5 x the row's bar of the k-th place (both selections are exact on their own
#   float32 values, each within the bar of the float64 one).
HEAVY_F64_BOUND = 2e-5
LIGHT_MACS = 1_000_000


def _check_against_float64(got, want, exact, k, what, macs=None, heavy=None, stats=None):
    """`exact[i]`: the float64 dense row of sampled row i.  Returns the observed maxima per row class (written to profiles/ by the caller)."""
    if stats is None:
        stats = {c: {"rows": 0, "hip_vs_f64": 0.0, "ref_vs_f64": 0.0, "hip_vs_ref": 0.0, "set_diff_rows": 0} for c in ("heavy", "light_le_1e6_macs", "other")}
    for i, ((gc, gv), (wc, wv)) in enumerate(zip(got, want)):
        assert gc.shape[0] == wc.shape[0], f"{what}: slot {i}: kept {gc.shape[0]} entries, expected {wc.shape[0]}"
        if gc.shape[0] == 0:
            continue
        e = exact[i]
        ref_err = np.abs(wv.astype(np.float64) - e[wc]) / np.abs(e[wc])
        hip_err = np.abs(gv.astype(np.float64) - e[gc]) / np.abs(e[gc])
        light = macs is not None and macs[i] <= LIGHT_MACS
        bar = RTOL if light else HEAVY_F64_BOUND
        cls = "heavy" if (heavy is not None and heavy[i]) else ("light_le_1e6_macs" if light else "other")
        st = stats[cls]
        st["rows"] += 1
        st["hip_vs_f64"] = max(st["hip_vs_f64"], float(hip_err.max()))
        st["ref_vs_f64"] = max(st["ref_vs_f64"], float(ref_err.max()))
        _, gi, wi = np.intersect1d(gc, wc, assume_unique=True, return_indices=True)
        if gi.size:
            st["hip_vs_ref"] = max(st["hip_vs_ref"], float(np.max(np.abs(gv[gi].astype(np.float64) - wv[wi]) / np.abs(wv[wi].astype(np.float64)))))
        assert hip_err.max() <= bar, (f"{what}: slot {i} ({macs[i] if macs is not None else '?'} MACs): HIP values up to {hip_err.max():.2e} from the float64 value "
                                      f"(bar {bar:.1e}; the reference port {ref_err.max():.2e})")
        if light and gi.size:
            np.testing.assert_allclose(gv[gi], wv[wi], rtol=RTOL, atol=ATOL, err_msg=f"{what}: slot {i} ({macs[i]} MACs) vs the reference")
        g_only, w_only = np.setdiff1d(gc, wc), np.setdiff1d(wc, gc)
        if g_only.size:
            st["set_diff_rows"] += 1
            assert gc.shape[0] == k, f"{what}: slot {i}: different columns although fewer than k were kept"
            kth = min(e[gc].min(), e[wc].min())                         # the float64 value at the k-th place
            tie = 2.5 * max(bar, float(ref_err.max()))                  # (the reference's side of the tie carries the reference's own error)
            for c in np.concatenate((g_only, w_only)):
                assert abs(e[c] - kth) <= tie * abs(kth), f"{what}: slot {i}: column {c} (value {e[c]}) is not on the k-th place tie ({kth})"
    return stats


def _slots_from_csr(res: sp.csr_array, rows):
    out = []
    for t in rows:
        c = res.indices[res.indptr[t]:res.indptr[t + 1]]
        v = res.data[res.indptr[t]:res.indptr[t + 1]]
        o = np.argsort(c, kind="stable")
        out.append((c[o].astype(np.int32), v[o].astype(np.float32)))
    return out


def _slots_from_flat(cols, vals, counts, k, rows):
    out = []
    for t in rows:
        n = int(counts[t])
        c, v = cols[t * k: t * k + n], vals[t * k: t * k + n]
        o = np.argsort(c, kind="stable")
        out.append((c[o], v[o]))
    return out


def _oracle_slots(call, sample, drop_zeros=False):
    sub = copy.copy(call)
    sub.targets = np.ascontiguousarray(sample, dtype=np.int32)
    want = so.canonical(*so.run_kernel(sub, "port"), sub.targets, call.k)
    if drop_zeros:      # CSR output: genuine zeros are removed (s_plus.pyx:424)
        want = [(c[v != 0], v[v != 0]) for c, v in want]
    return want


# ------------------------------------------------------------------------------------------------------------
# configs[0]
# ------------------------------------------------------------------------------------------------------------
def test_config0_cosine_sps_random_all_rows():
    m = workloads.c1_matrix()
    assert m.shape == (10_000, 20_000)
    call = _host.prepare(m, k=50, l2=1, c1=0.5, c2=0.5)
    rows, cols, vals, counts = _host.run_hip(call)
    got = so.canonical(rows, cols, vals, call.targets, 50)
    want = so.canonical(*so.run_kernel(call, "port"), call.targets, 50)
    so.compare_topk(got, want, 50, rtol=RTOL, atol=ATOL, what="C1")
    # ... and through the public wrapper, the reference tests' comparator (check_sum, tests/test_similarity.py:8-14)
    res = sim.cosine(m, k=50, verbose=False, format_output="csr")
    assert res.shape == (10_000, 10_000) and res.dtype == np.float32
    ref = sp.csr_array((np.concatenate([v for _, v in want]), np.concatenate([c for c, _ in want]),
                        np.concatenate(([0], np.cumsum([c.shape[0] for c, _ in want])))), shape=res.shape)
    cs = lambda x: float(np.sum(np.asarray(x.sum(axis=1), dtype=np.float64).ravel() ** 2))  # noqa: E731
    np.testing.assert_allclose(cs(res), cs(ref), rtol=1e-4)


# ------------------------------------------------------------------------------------------------------------
# configs[1] and configs[2]: one matrix
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2_matrix():
    return workloads.fixed_degree_csr(1_000_000, 100_000, 64, 12345)


@pytest.fixture(scope="module")
def c2(c2_matrix):
    m = c2_matrix
    call = _host.prepare(m, k=100, l2=1, c1=0.5, c2=0.5)
    rows, cols, vals, counts, info = _host.run_hip(call, time_kernel=True)
    return m, call, rows.reshape(-1, 100), cols.reshape(-1, 100), vals.reshape(-1, 100), counts, info


def test_config1_full_size_properties(c2):
    m, call, rows, cols, vals, counts, info = c2
    n, k = 1_000_000, 100
    # every row has far more than k candidates (~40k): all slots are used, no padding
    assert counts.min() == k and counts.max() == k
    assert np.array_equal(rows, np.broadcast_to(np.arange(n, dtype=np.int32)[:, None], (n, k)))
    assert cols.min() >= 0 and cols.max() < n
    # cosine of non-negative data: 0 < value <= 1 (+ float32 slack)
    assert vals.min() > 0.0 and vals.max() <= 1.0 + 2e-6
    # the row itself is always among its neighbours with similarity 1 (diagonal is kept)
    self_pos = (cols == np.arange(n, dtype=np.int32)[:, None])
    assert self_pos.sum(axis=1).min() == 1 and self_pos.sum(axis=1).max() == 1
    np.testing.assert_allclose(vals[self_pos], 1.0, rtol=1e-5)
    assert np.array_equal(vals.max(axis=1), vals[self_pos])
    # no column twice in a slot (spot check on 20k rows: a full check would sort 1e8 entries)
    pick = np.random.default_rng(0).choice(n, 20_000, replace=False)
    srt = np.sort(cols[pick], axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])
    # symmetry of cosine on m @ m.T: if j is in i's list with value v and i is in j's list, the values agree
    for i in pick[:200]:
        for j, v in zip(cols[i, :5], vals[i, :5]):
            hit = np.flatnonzero(cols[j] == i)
            if hit.size:
                assert abs(vals[j, hit[0]] - v) <= 1e-5 * max(abs(v), 1e-12)
    # the headline shape runs on the sparse (bitmap) kernel, nothing handed to the generic one
    ph = info["phase_cycles"]
    assert ph[9] == n and ph[10] == 0, (ph[9], ph[10])


def test_config1_sample_vs_oracle(c2):
    m, call, rows, cols, vals, counts, info = c2
    k = 100
    sample = _sample_with_queue_tail(_macs_per_row(m, call.m2_indptr), 1)
    assert sample.shape[0] >= N_SAMPLE
    want = _oracle_slots(call, sample)
    got = _slots_from_flat(cols.ravel(), vals.ravel(), counts, k, sample)
    so.compare_topk(got, want, k, rtol=RTOL, atol=ATOL, what="C2 sample")


def test_config2_splus_hybrid_full_size(c2_matrix):
    """s_plus (tversky + cosine hybrid, stabilized shrink 10) on the 1M x 100k matrix: the GENERAL variant of the sparse
    kernel (survivor pool + judge) at its defining shape."""
    m = c2_matrix
    n, k = 1_000_000, 100
    call = _host.prepare(m, k=k, l1=0.5, l2=0.5, stabilized_shrink=10.0)
    rows, cols, vals, counts, info = _host.run_hip(call, time_kernel=True)
    ph = info["phase_cycles"]
    assert ph[9] == n and ph[10] == 0, f"rows on the sparse kernel {ph[9]}, handed to the generic kernel {ph[10]}"
    assert counts.min() == k and counts.max() == k
    c2d, v2d = cols.reshape(n, k), vals.reshape(n, k)
    assert c2d.min() >= 0 and c2d.max() < n and np.isfinite(v2d).all() and v2d.min() > 0
    # hybrid of non-negative data with shrink: every value is below the unshrunk cosine's maximum of 1
    assert v2d.max() < 1.0
    # the row itself: xy = |x|^2, den = .5*|x|^2 + .5*|x|^2 + 10  ->  |x|^2 / (|x|^2 + 10), and it is the row's maximum
    sq = np.add.reduceat(np.square(m.data, dtype=np.float32), m.indptr[:-1])
    self_pos = (c2d == np.arange(n, dtype=np.int32)[:, None])
    assert self_pos.sum(axis=1).min() == 1
    np.testing.assert_allclose(v2d[self_pos], sq / (sq + np.float32(10.0)), rtol=2e-5)
    sample = _sample_with_queue_tail(_macs_per_row(m, call.m2_indptr), 2)
    assert sample.shape[0] >= N_SAMPLE
    want = _oracle_slots(call, sample)
    got = _slots_from_flat(cols, vals, counts, k, sample)
    so.compare_topk(got, want, k, rtol=RTOL, atol=ATOL, what="C3 sample")


# ------------------------------------------------------------------------------------------------------------
# configs[3]: p3alpha + rp3beta on the MovieLens-32M-shaped URM, item-item, k=200, public wrappers
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c4():
    urm = workloads.movielens_like_urm()
    assert urm.shape == (200_948, 84_432) and urm.nnz == 32_000_204
    m1 = urm.T.tocsr()
    m1.sort_indices()
    # the oracle's operands: the reference's own preprocessing (similarity.py:410-415, 477-483) in NumPy
    m2 = m1.T.tocsr()
    pop_m2 = np.asarray(m2.sum(axis=0)).ravel()
    a = normalize(m1, norm="l1", axis=1)
    a.data = np.power(a.data, 0.8)
    b = normalize(m2, norm="l1", axis=1)
    b.data = np.power(b.data, 0.8)
    # sample: the 20 heaviest rows (most MACs: popular items, they reach nearly every column) + 5 000 random ones (320 rows until round 5)
    nnz2 = np.diff(b.indptr).astype(np.int64)
    per = nnz2[a.indices]
    csum = np.concatenate(([0], np.cumsum(per)))
    macs = csum[a.indptr[1:]] - csum[a.indptr[:-1]]
    heavy = np.argsort(-macs)[:20]
    rnd = np.random.default_rng(4).choice(m1.shape[0], 5000, replace=False)
    sample = np.unique(np.concatenate((heavy, rnd))).astype(np.int32)
    return m1, a, b, pop_m2, sample, macs


@pytest.mark.parametrize("name", ["p3alpha", "rp3beta"])
def test_config3_p3_item_item_public_wrappers(c4, name):
    m1, a, b, pop_m2, sample, macs = c4
    k = 200
    if name == "p3alpha":
        res = sim.p3alpha(m1, alpha=0.8, k=k, verbose=False, format_output="csr")
        call = _host.prepare(a, b, k=k, target_rows=sample)
    else:
        res = sim.rp3beta(m1, alpha=0.8, beta=0.4, k=k, verbose=False, format_output="csr")
        call = _host.prepare(a, b, k=k, weight_depop_matrix2=pop_m2, p2=0.4, l3=1, target_rows=sample)
    n = m1.shape[0]
    assert res.shape == (n, n) and res.dtype == np.float32 and isinstance(res, sp.csr_array)
    row_nnz = np.diff(res.indptr)
    assert row_nnz.max() <= k and res.indices.min() >= 0 and res.indices.max() < n
    assert np.isfinite(res.data).all() and res.data.min() > 0
    # items nobody rated have no neighbours; every rated item reaches at least itself
    rated = np.diff(m1.indptr) > 0
    assert not row_nnz[~rated].any() and row_nnz[rated].min() >= 1
    want = _oracle_slots(call, sample, drop_zeros=True)
    got = _slots_from_csr(res, sample)
    # float64 statement of the sampled rows: xy = a[t] . b[:, c]; p3alpha returns it, rp3beta divides by l3 * Xdepop[t] * Ydepop[c]
    # with the float32 column terms the kernel is handed (s_plus.h:129-156)
    # (the dense float64 judge in chunks of 256 rows: 256 x 84 432 x 8 B at a time)
    a64, b64 = sp.csr_array(a, dtype=np.float64), sp.csr_array(b, dtype=np.float64)
    heavy_set = set(np.argsort(-macs)[:20].tolist())
    stats = None
    for c0 in range(0, len(sample), 256):
        sl = slice(c0, min(len(sample), c0 + 256))
        exact = (a64[sample[sl]] @ b64).toarray()
        if name == "rp3beta":
            exact = exact / (call.Xdepop.astype(np.float64)[sample[sl], None] * call.Ydepop.astype(np.float64)[None, :])
        stats = _check_against_float64(got[sl], want[sl], exact, k, f"C4 {name}", macs=macs[sample[sl]], heavy=[int(t) in heavy_set for t in sample[sl]], stats=stats)
        del exact
    # the observed maxima, for profiles/r06_c4_value_errors.txt (the GPU box merges gpurun_out/ back)
    out_dir = ROOT / "gpurun_out"
    out_dir.mkdir(exist_ok=True)
    with open(out_dir / f"c4_value_errors_{name}.txt", "w") as f:
        f.write(f"configs[3] {name}, MovieLens-32M-SHAPED stand-in URM (workloads.movielens_like_urm, seed 0: no network for the real file), k={k}, "
                f"{len(sample)} sampled rows; relative errors of kept values, maxima per row class\n")
        f.write("class                rows  HIP vs float64   reference vs float64   HIP vs reference   rows whose column sets differ (k-th place ties)\n")
        for cls, st in stats.items():
            f.write(f"{cls:20s} {st['rows']:4d}  {st['hip_vs_f64']:.3e}        {st['ref_vs_f64']:.3e}              {st['hip_vs_ref']:.3e}          {st['set_diff_rows']}\n")
        f.write(f"bar (asserted): rows of <= 1e6 MACs: HIP within 1e-5 of the float64 value and of the reference; heavier rows: HIP within {HEAVY_F64_BOUND:.1e} of the "
                "float64 value (float32 sums of 1e5 .. 1e6 products: the reference itself is up to 1.4e-5 from it)\n")


# ------------------------------------------------------------------------------------------------------------
# configs[4]: user scoring with filter_cols=urm, one GPU's 1M-user slice of the 10M-user job
# ------------------------------------------------------------------------------------------------------------
def test_config4_user_scoring_with_seen_filter():
    n_users, n_items, k = 1_000_000, 100_000, 100
    urm = workloads.fixed_degree_csr(n_users, n_items, 64, 12345)
    W = sim.cosine(urm[:200_000].T.tocsr(), k=100, verbose=False, format_output="csr")      # item-item model from a user subsample
    assert W.shape == (n_items, n_items)
    Wt = W.T.tocsr()
    res = sim.dot_product(urm, Wt, k=k, filter_cols=urm, verbose=False, format_output="csr")
    assert res.shape == (n_users, n_items) and res.dtype == np.float32
    row_nnz = np.diff(res.indptr)
    assert row_nnz.max() <= k and row_nnz.min() > 0
    # nothing seen is recommended — over ALL rows (pattern product in chunks of 100k users)
    pat = sp.csr_array((np.ones(urm.nnz, dtype=np.float32), urm.indices, urm.indptr), shape=urm.shape)
    for lo in range(0, n_users, 100_000):
        hi = lo + 100_000
        assert res[lo:hi].multiply(pat[lo:hi]).nnz == 0, f"users {lo}..{hi}: a seen item was recommended"
    # scores of non-negative data are positive and sorted sets are duplicate-free (spot check)
    assert res.data.min() > 0
    call = _host.prepare(urm, Wt, k=k, filter_cols=urm)
    sample = _sample_with_queue_tail(_macs_per_row(urm, Wt.indptr), 5)
    assert sample.shape[0] >= N_SAMPLE
    want = _oracle_slots(call, sample, drop_zeros=True)
    got = _slots_from_csr(res, sample)
    so.compare_topk(got, want, k, rtol=RTOL, atol=ATOL, what="C5 sample")


# ------------------------------------------------------------------------------------------------------------
# beyond every config: nnz(matrix2) >= 2^30 (the reference's limit is 2^31 - 1, s_plus.pyx:241-244)
# ------------------------------------------------------------------------------------------------------------
def test_matrix2_with_more_than_2_30_entries():
    """4.3 GB per m2 stream: past what a 32-bit byte offset reaches.  The library sends such calls to the generic kernel's
    64-bit-offset variant.  m2: 1 050 000 rows x 1024 entries (columns ascending inside a row, values from a small
    set); the targets' m1 rows point at m2 rows over the whole range, the last ones beyond the 4 GB mark."""
    R, L, n_cols, k = 1_050_000, 1024, 100_000, 50
    assert R * L >= 2 ** 30
    r = np.arange(R, dtype=np.int32)[:, None]
    j = np.arange(L, dtype=np.int32)[None, :]
    indices = (j * np.int32(97) + (r * np.int32(7)) % np.int32(672)).astype(np.int32, copy=False).ravel()      # < 99 328 + 672 = n_cols
    data = (((r + j) % np.int32(13)) + np.int32(1)).astype(np.float32).ravel()
    data *= np.float32(0.125)
    indptr = np.arange(R + 1, dtype=np.int64) * L
    assert indptr[-1] < 2 ** 31
    m2 = sp.csr_array((data, indices, indptr.astype(np.int32)), shape=(R, n_cols))
    rng = np.random.default_rng(3)
    n_t, per = 48, 40
    picks = [np.unique(np.concatenate([rng.choice(R - 2000, per - 8), rng.integers(R - 1400, R, 8)])) for _ in range(n_t)]   # ascending, duplicate-free
    lens = np.array([p.size for p in picks])
    u_all = np.concatenate(picks).astype(np.int32)
    m1 = sp.csr_array((rng.random(u_all.size, dtype=np.float32) + np.float32(0.5), u_all,
                       np.concatenate(([0], np.cumsum(lens))).astype(np.int32)), shape=(n_t, R))
    assert all(int(p.max()) * L * 4 >= 2 ** 32 for p in picks), "every target row reads m2 beyond the 4 GB mark"
    call = _host.prepare(m1, m2, k=k)
    rows, cols, vals, counts = _host.run_hip(call)
    got = so.canonical(rows, cols, vals, call.targets, k)
    want = so.canonical(*so.run_kernel(call, "port"), call.targets, k)
    so.compare_topk(got, want, k, rtol=RTOL, atol=ATOL, what="nnz(m2) >= 2^30")
    assert counts.min() == k


def test_config4_ten_million_users_streamed_in_chunks():
    """BASELINE configs[4] at its defining size on ONE GPU: dot_product(urm, W.T, k=100, filter_cols=urm) for 10 M users x 100 k
    items through the chunked route (`multi_gpu.similarity(..., chunk_rows=1_000_000)`: the operands resident once, the target
    list streamed, every chunk turned into its CSR rows on arrival — README.md:88-94, tests/test_similarity.py:543-615 of the
    reference at 10^4 times their size).  Checked: shape and row lengths, NOTHING a user has seen is recommended (all 10^9
    entries, looked up on the GPU), >= 300 sampled rows against the oracle — first and last rows and both sides of every chunk
    boundary included —, and the parent's peak memory stays below what the unchunked assembly would need."""
    import resource
    import torch

    n_chunk, n_items, k, chunk = 10, 100_000, 100, 1_000_000
    n_users = n_chunk * chunk
    urm = sp.vstack([workloads.fixed_degree_csr(chunk, n_items, 64, 500 + i) for i in range(n_chunk)], format="csr")
    assert urm.shape == (n_users, n_items) and urm.indices.dtype == np.int32
    W = sim.cosine(urm[:200_000].T.tocsr(), k=100, verbose=False, format_output="csr")      # item-item model from a user subsample
    Wt = W.T.tocsr()
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
    res = sim.multi_gpu.similarity("dot_product", urm, Wt, k=k, filter_cols=urm, devices=[0], chunk_rows=chunk, format_output="csr", verbose=False)
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024
    assert res.shape == (n_users, n_items) and res.dtype == np.float32 and isinstance(res, sp.csr_array)
    row_nnz = np.diff(res.indptr)
    assert row_nnz.max() <= k and row_nnz.min() > 0 and res.data.min() > 0
    # the unchunked route holds rows / cols / values of all n * k slots (12 bytes each) next to the CSR it builds from them;
    # the streamed one only ever holds the CSR pieces (8 bytes per kept entry) and their concatenation
    csr_bytes = 8 * res.nnz
    assert rss1 - rss0 < 2 * csr_bytes + 4 * 2 ** 30 < 2 * csr_bytes + 12 * n_users * k, (rss0, rss1, csr_bytes)
    # nothing seen is recommended: every (user, item) of the result looked up in the (sorted) keys of the URM, on the GPU
    dev = torch.device("cuda", 0)
    seen = (torch.arange(n_users, device=dev, dtype=torch.int64).repeat_interleave(torch.from_numpy(np.diff(urm.indptr).astype(np.int64)).to(dev)) * n_items
            + torch.from_numpy(urm.indices).to(dev))
    assert bool((seen[1:] > seen[:-1]).all())
    for lo in range(0, n_users, chunk):
        p0, p1 = int(res.indptr[lo]), int(res.indptr[lo + chunk])
        users = torch.arange(lo, lo + chunk, device=dev, dtype=torch.int64).repeat_interleave(torch.from_numpy(row_nnz[lo:lo + chunk].astype(np.int64)).to(dev))
        keys = users * n_items + torch.from_numpy(res.indices[p0:p1].astype(np.int64)).to(dev)
        pos = torch.searchsorted(seen, keys).clamp_(max=seen.numel() - 1)
        assert not bool((seen[pos] == keys).any()), f"users {lo}..{lo + chunk}: a seen item was recommended"
    del seen
    torch.cuda.empty_cache()
    # the oracle on a sample: both ends, both sides of every chunk boundary, random rows of every chunk
    rng = np.random.default_rng(6)
    edge = np.concatenate([[0, 1, n_users - 2, n_users - 1]] + [[b - 2, b - 1, b, b + 1] for b in range(chunk, n_users, chunk)])
    sample = np.unique(np.concatenate((edge, rng.choice(n_users, 280, replace=False)))).astype(np.int64)
    assert sample.shape[0] >= 300
    sub = urm[sample]
    call = _host.prepare(sub, Wt, k=k, filter_cols=sub)
    want = _oracle_slots(call, np.arange(sample.shape[0]), drop_zeros=True)
    got = _slots_from_csr(res, sample)
    so.compare_topk(got, want, k, rtol=RTOL, atol=ATOL, what="configs[4], 10 M users")
