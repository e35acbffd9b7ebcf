/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.
 *
 * Plain-C (C99 + OpenMP) restatement of the one compute kernel of
 * bogliosimone/similaripy: s_plus::compute_similarities_parallel<int,float>
 * (reference: similaripy/cython_code/s_plus.h:265-453) together with its two
 * helpers TopK (s_plus.h:39-64) and SparseMatrixMultiplier (s_plus.h:71-240).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  similaripy_amd/ never does: the product path is the HIP
 * library and fails loudly when that is missing.
 *
 * Parity: PINNED.  tests/test_oracle.py checks this port (a) bit-for-bit
 * against oracle/_ref/libsplus_ref.so, which is the reference header itself
 * compiled in place from /root/reference (see oracle/Makefile), and (b) against
 * the golden vectors in tests/golden/ that were produced by importing the
 * reference Python package (tests/golden/make_golden.py).
 *
 * Written from the behaviour of the reference, not copied from it: the
 * accumulator is a dense float array plus a first-touch list, the heap is a
 * hand-rolled binary min-heap on (score, index) pairs with the same ordering
 * std::greater<std::pair<float,int>> gives the reference (ties on score are
 * broken by index, smallest pair at the root).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SEL_NONE 0
#define SEL_ARRAY 1
#define SEL_MATRIX 2

typedef struct { float score; int index; } pair_t;

/* (a.score, a.index) < (b.score, b.index) lexicographically */
static inline int pair_less(pair_t a, pair_t b) {
    if (a.score < b.score) return 1;
    if (b.score < a.score) return 0;
    return a.index < b.index;
}

/* Bounded min-heap, root = smallest pair.  s_plus.h:39-64. */
typedef struct { pair_t *h; int n; int k; } topk_t;

static void heap_sift_up(pair_t *h, int i) {
    pair_t v = h[i];
    while (i > 0) {
        int p = (i - 1) >> 1;
        if (!pair_less(v, h[p])) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = v;
}

static void heap_sift_down(pair_t *h, int n, int i) {
    pair_t v = h[i];
    for (;;) {
        int c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && pair_less(h[c + 1], h[c])) c++;
        if (!pair_less(h[c], v)) break;
        h[i] = h[c];
        i = c;
    }
    h[i] = v;
}

/* s_plus.h:45-59: fill to k, then replace the root iff score > root.score (strict). */
static inline void topk_offer(topk_t *t, int index, float score) {
    if (t->n < t->k) {
        t->h[t->n].score = score;
        t->h[t->n].index = index;
        heap_sift_up(t->h, t->n);
        t->n++;
    } else if (score > t->h[0].score) {
        t->h[0].score = score;
        t->h[0].index = index;
        heap_sift_down(t->h, t->n, 0);
    }
}

/* first position in sorted a[lo,hi) with a[pos] >= x (std::lower_bound, s_plus.h:390-394) */
static inline int lower_bound_i32(const int *a, int lo, int hi, int x) {
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* sorted-range membership, std::binary_search equivalent (s_plus.h:165-169, 181-185) */
static inline int range_has(const int *a, int lo, int hi, int x) {
    int p = lower_bound_i32(a, lo, hi, x);
    return p < hi && a[p] == x;
}

typedef struct {
    const float *Xtv, *Ytv, *Xcos, *Ycos, *Xdep, *Ydep;
    float a1, l1, l2, l3, t1, t2, stab, bayes, threshold;
    int filter_mode; const int *f_indptr, *f_indices;
    int target_mode; const int *t_indptr, *t_indices;
} epi_t;

/* s_plus.h:129-156.  Tversky term uses the RAW xy; pow only when a1 != 1;
 * when no normalisation/shrink is active the raw dot is returned. */
static inline float epilogue(const epi_t *e, int row, int col, float xy) {
    float vt = 0.f, vc = 0.f, vd = 0.f, val = xy;
    if (e->l1 != 0.f) vt = e->l1 * (e->t1 * (e->Xtv[row] - xy) + e->t2 * (e->Ytv[col] - xy) + xy);
    if (e->l2 != 0.f) vc = e->l2 * (e->Xcos[row] * e->Ycos[col]);
    if (e->l3 != 0.f) vd = e->l3 * (e->Xdep[row] * e->Ydep[col]);
    if (e->a1 != 1.f) xy = powf(xy, e->a1);
    if (e->l1 != 0.f || e->l2 != 0.f || e->l3 != 0.f || e->stab != 0.f || e->bayes != 0.f) {
        float den = vt + vc + vd + e->stab;
        val = (den != 0.f) ? xy / den : 0.f;
        if (e->bayes != 0.f) val = val * (xy / (xy + e->bayes));
    }
    return val;
}

/* s_plus.h:192-215: visit first-touch list, apply selectors/epilogue/threshold, clear. */
static void drain(float *sums, int *touched, int *n_touched, int block_offset,
                  int row, const epi_t *e, topk_t *tk) {
    int n = *n_touched;
    for (int i = 0; i < n; ++i) {
        int lc = touched[i];
        float xy = sums[lc];
        int col = block_offset + lc;
        int filtered = 0, targeted = 1;
        if (e->filter_mode == SEL_MATRIX)
            filtered = range_has(e->f_indices, e->f_indptr[row], e->f_indptr[row + 1], col);
        if (!filtered && e->target_mode == SEL_MATRIX)
            targeted = range_has(e->t_indices, e->t_indptr[row], e->t_indptr[row + 1], col);
        if (!filtered && targeted) {
            float val = epilogue(e, row, col, xy);
            if (val >= e->threshold) topk_offer(tk, col, val);
        }
        sums[lc] = 0.f;
    }
    *n_touched = 0;
}

/* s_plus.h:112-117: first touch is detected by sums == 0 (so a partial sum that
 * cancels to exactly 0 is re-listed — kept on purpose). */
#define ACC_ADD(c, v) do { if (sums[(c)] == 0.f) { \
        if (n_touched == cap_touched) { cap_touched *= 2; touched = (int*)realloc(touched, sizeof(int) * (size_t)cap_touched); } \
        touched[n_touched++] = (c); } sums[(c)] += (v); } while (0)

void splus_port_compute(
    int n_targets, const int *targets,
    const float *m1_data, const int *m1_indices, const int *m1_indptr,
    const float *m2_data, const int *m2_indices, const int *m2_indptr,
    const float *Xtv, const float *Ytv, const float *Xcos, const float *Ycos,
    const float *Xdep, const float *Ydep,
    float a1, float l1, float l2, float l3, float t1, float t2,
    float stab, float bayes, float threshold,
    int k, int n_output_cols,
    int filter_mode, const int *f_indptr, const int *f_indices,
    int target_mode, const int *t_indptr, const int *t_indices,
    int *rows, int *cols, float *values,
    int num_threads, int block_size)
{
    /* s_plus.h:309-311 */
    const int n_blocks = (block_size > 0) ? (n_output_cols + block_size - 1) / block_size : 1;
    const int use_blocking = (block_size > 0) && (n_output_cols > block_size);
    const int acc_len = (block_size > 0 && block_size < n_output_cols) ? block_size : n_output_cols;

    epi_t e;
    e.Xtv = Xtv; e.Ytv = Ytv; e.Xcos = Xcos; e.Ycos = Ycos; e.Xdep = Xdep; e.Ydep = Ydep;
    e.a1 = a1; e.l1 = l1; e.l2 = l2; e.l3 = l3; e.t1 = t1; e.t2 = t2;
    e.stab = stab; e.bayes = bayes; e.threshold = threshold;
    e.filter_mode = filter_mode; e.f_indptr = f_indptr; e.f_indices = f_indices;
    e.target_mode = target_mode; e.t_indptr = t_indptr; e.t_indices = t_indices;

#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#else
    num_threads = 1;
#endif

#pragma omp parallel num_threads(num_threads)
    {
        float *sums = (float *)calloc((size_t)(acc_len > 0 ? acc_len : 1), sizeof(float));
        int cap_touched = 1024, n_touched = 0;
        int *touched = (int *)malloc(sizeof(int) * (size_t)cap_touched);
        topk_t tk;
        tk.k = k; tk.n = 0;
        tk.h = (pair_t *)malloc(sizeof(pair_t) * (size_t)(k > 0 ? k : 1));

#pragma omp for schedule(dynamic)
        for (int i = 0; i < n_targets; ++i) {
            const int t = targets[i];
            const int s1 = m1_indptr[t], e1 = m1_indptr[t + 1];
            tk.n = 0;

            if (use_blocking) {
                /* s_plus.h:350-410: one accumulate+drain per column block; heap persists */
                for (int b = 0; b < n_blocks; ++b) {
                    const int cb0 = b * block_size;
                    const int cb1 = (cb0 + block_size < n_output_cols) ? cb0 + block_size : n_output_cols;
                    for (int p = s1; p < e1; ++p) {
                        const int u = m1_indices[p];
                        const float v1 = m1_data[p];
                        int lo = m2_indptr[u], hi = m2_indptr[u + 1];
                        if (lo == hi) continue;
                        if (m2_indices[hi - 1] < cb0 || m2_indices[lo] >= cb1) continue;
                        if (m2_indices[lo] < cb0) lo = lower_bound_i32(m2_indices, lo, hi, cb0);
                        if (m2_indices[hi - 1] >= cb1) hi = lower_bound_i32(m2_indices, lo, hi, cb1);
                        for (int q = lo; q < hi; ++q) {
                            const int lc = m2_indices[q] - cb0;
                            const float pv = v1 * m2_data[q];
                            ACC_ADD(lc, pv);
                        }
                    }
                    if (n_touched > 0) drain(sums, touched, &n_touched, cb0, t, &e, &tk);
                }
            } else {
                /* s_plus.h:411-441 */
                for (int p = s1; p < e1; ++p) {
                    const int u = m1_indices[p];
                    const float v1 = m1_data[p];
                    const int lo = m2_indptr[u], hi = m2_indptr[u + 1];
                    for (int q = lo; q < hi; ++q) {
                        const int c = m2_indices[q];
                        const float pv = m2_data[q] * v1;
                        ACC_ADD(c, pv);
                    }
                }
                drain(sums, touched, &n_touched, 0, t, &e, &tk);
            }

            /* s_plus.h:444-450: heap array order, slot i owns [k*i, k*i+k), tail untouched */
            long long o = (long long)k * (long long)i;
            for (int j = 0; j < tk.n; ++j) {
                rows[o + j] = t;
                cols[o + j] = tk.h[j].index;
                values[o + j] = tk.h[j].score;
            }
        }
        free(sums); free(touched); free(tk.h);
    }
}

int splus_port_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
