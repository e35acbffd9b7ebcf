// sp_knn.hip — MI355X (gfx950 / CDNA4) top-k sparse row similarity.
//
// One kernel replaces the reference's only native hot path,
//   s_plus::compute_similarities_parallel<int,float>   (similaripy/cython_code/s_plus.h:265-453)
// i.e. for every target row t of CSR m1:
//   acc[c] = sum_u m1[t,u] * m2[u,c]            (Gustavson row-wise SpGEMM, s_plus.h:418-438)
//   val[c] = epilogue(acc[c], X*[t], Y*[c])     (s_plus.h:129-156)
//   keep the k largest val[c] >= threshold that pass the column selectors (s_plus.h:192-215, 39-64)
//
// MI355X mapping:
//   * one persistent 64-lane-wave workgroup per CU slot pulls target rows from a queue
//     (the analogue of `omp for schedule(dynamic)`, s_plus.h:337);
//   * the per-thread dense `sums[]` array of the reference (n_cols*4 B, cache-hostile) becomes an
//     LDS-resident accumulator tile of T slots fed with ds_add_f32 / ds_cmpst atomics:
//       - direct-indexed ("dense") when the current column window is <= T columns,
//       - open-addressing hash (multiplicative hash, linear probing) otherwise;
//     rows whose candidates do not fit are processed in several column windows, exactly the
//     reference's blocked path (s_plus.h:350-410: window = [cb_start, cb_end), sub-range of each
//     sorted m2 row found by lower_bound), with the top-k state carried across windows;
//   * m2 rows are streamed with lane-contiguous (coalesced) index/value loads: the nnz1(t)
//     segments of a window are flattened through an LDS prefix array so all 64 lanes stay busy
//     whatever the segment lengths;
//   * the std::push_heap/pop_heap TopK becomes a workgroup-wide selection: survivors of a running
//     threshold are appended to an LDS candidate buffer and, when it fills, an MSD radix-select
//     (4 x 8-bit passes over an order-preserving key) keeps exactly k.
// HBM-bound integer/float streaming work: no MFMA on purpose.
//
// Everything below is written for gfx950 only (wave64, 160 KiB LDS, ds_add_f32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <algorithm>
#include <vector>

#include "../../include/sp_knn.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int EMPTY = -1;        // key of a free accumulator slot (column ids are >= 0)
constexpr int DRAIN_UNROLL = 2;  // slots per thread between two capacity checks of the candidate buffer
constexpr int MAX_PROBE = 128;   // linear-probe budget before a window is declared overflowed

// scalar slots in LDS
enum { SH_CNT = 0, SH_OVF, SH_SEL, SH_NEED, SH_EQ, SH_CNT2, SH_NEXT, SH_TOTAL, SH_N };

struct KParams {
    int n_targets;
    const int *targets;
    const float *m1_data; const int *m1_indices; const int *m1_indptr;
    const float *m2_data; const int *m2_indices; const int *m2_indptr;
    const float *Xtv, *Ytv, *Xcos, *Ycos, *Xdep, *Ydep;
    float a1, l1, l2, l3, t1, t2, stab, bayes, threshold;
    int k;
    int n_cols;
    int filter_mode; const int *f_indptr; const int *f_indices;
    int target_mode; const int *t_indptr; const int *t_indices;
    int *rows; int *cols; float *values; int *counts;
    // configuration
    int T;                 // accumulator slots (power of two)
    int logT;
    int cap;               // candidate buffer capacity (>= k + NT*DRAIN_UNROLL)
    u64 *gU;               // candidate buffers in global memory (only when they do not fit LDS)
    unsigned int *queue;   // [0] = next slot index (dynamic scheduling), [1] = pass counter (debug)
    const int *order;      // optional: slot visiting order (descending work); NULL = identity
    int hash_fill;         // slots' worth of MACs one hash window may receive (= T * load_pct / 100)
    int static_sched;
    int count_passes;
};

// order-preserving float <-> uint map (so radix-select works for negative thresholds too)
__device__ __forceinline__ unsigned fkey(float f) {
    unsigned b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ int lower_bound_g(const int *__restrict__ a, int lo, int hi, int x) {
    while (lo < hi) {
        int mid = lo + ((hi - lo) >> 1);
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ bool range_has(const int *__restrict__ a, int lo, int hi, int x) {
    int p = lower_bound_g(a, lo, hi, x);
    return p < hi && a[p] == x;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Epilogue of s_plus.h:129-156: Tversky uses the RAW xy, pow only if a1 != 1,
// raw dot returned when no normalisation/shrink is active, den == 0 -> 0.
struct Epi {
    float a1, l1, l2, l3, t1, t2, stab, bayes;
    float xtv, xcos, xdep;  // row terms
    const float *Ytv, *Ycos, *Ydep;
    bool any;
    __device__ __forceinline__ float operator()(int col, float xy) const {
        float vt = 0.f, vc = 0.f, vd = 0.f, val = xy;
        if (l1 != 0.f) vt = l1 * (t1 * (xtv - xy) + t2 * (Ytv[col] - xy) + xy);
        if (l2 != 0.f) vc = l2 * (xcos * Ycos[col]);
        if (l3 != 0.f) vd = l3 * (xdep * Ydep[col]);
        if (a1 != 1.f) xy = powf(xy, a1);
        if (any) {
            float den = vt + vc + vd + stab;
            val = (den != 0.f) ? xy / den : 0.f;
            if (bayes != 0.f) val = val * (xy / (xy + bayes));
        }
        return val;
    }
};

// Keep exactly the k largest of U[0..n) (n > k), in place.  MSD radix-select on the 32-bit key in
// the high half of each entry.  Must be entered by the whole workgroup right after a barrier.
template <int NT>
__device__ void compact_topk(u64 *U, int *hist, int *sh, int k, bool &have_thr, unsigned &thr_key) {
    const int tid = threadIdx.x;
    const int n = sh[SH_CNT];
    __syncthreads();     // nobody may append (and change SH_CNT) before everyone has read n
    if (n <= k) return;  // uniform

    unsigned prefix = 0;
    int need = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned hmask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < n; i += NT) {
            unsigned key = (unsigned)(U[i] >> 32);
            if ((key & hmask) == (prefix & hmask)) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        if (tid < 64) {
            // lane L owns bins 255-4L .. 252-4L, i.e. lanes ascend as digits descend
            const int b0 = 255 - 4 * tid;
            const int c0 = hist[b0], c1 = hist[b0 - 1], c2 = hist[b0 - 2], c3 = hist[b0 - 3];
            const int s = c0 + c1 + c2 + c3;
            const int incl = wave_incl_scan(s);
            const int excl = incl - s;
            if (excl < need && need <= incl) {
                int r = need - excl, d;
                if (r <= c0) { d = b0; }
                else if (r <= c0 + c1) { d = b0 - 1; r -= c0; }
                else if (r <= c0 + c1 + c2) { d = b0 - 2; r -= c0 + c1; }
                else { d = b0 - 3; r -= c0 + c1 + c2; }
                sh[SH_SEL] = d;
                sh[SH_NEED] = r;
            }
        }
        __syncthreads();
        prefix |= (unsigned)sh[SH_SEL] << shift;
        need = sh[SH_NEED];
    }
    // prefix = k-th largest key; `need` entries equal to it are kept, everything larger is kept.
    if (tid == 0) { sh[SH_CNT2] = 0; sh[SH_EQ] = 0; }
    __syncthreads();
    const int lane = tid & 63;
    for (int base = 0; base < n; base += NT) {
        const int i = base + tid;
        u64 it = 0;
        bool keep = false;
        if (i < n) {
            it = U[i];
            unsigned key = (unsigned)(it >> 32);
            if (key > prefix) keep = true;
            else if (key == prefix) keep = atomicAdd(&sh[SH_EQ], 1) < need;
        }
        __syncthreads();  // every read of this chunk precedes the writes below (dest <= src index)
        const u64 m = __ballot(keep);
        if (m) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&sh[SH_CNT2], __popcll(m));
            wbase = __shfl(wbase, 0, 64);
            if (keep) U[wbase + __popcll(m & ((1ull << lane) - 1ull))] = it;
        }
    }
    __syncthreads();
    if (tid == 0) sh[SH_CNT] = sh[SH_CNT2];
    __syncthreads();
    have_thr = true;
    thr_key = prefix;
}

template <int NT, bool U_LDS>
__global__ __launch_bounds__(NT) void sp_knn_rows_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int T = p.T;

    // ---- LDS carve-up (single dynamic array; everything 8-byte aligned) ----
    int *keys = (int *)smem;                    // [T]
    float *vals = (float *)(keys + T);          // [T]
    int *seg_lo = (int *)(vals + T);            // [NT]   start of the window's slice of m2 row u
    int *seg_pre = seg_lo + NT;                 // [NT+64] exclusive prefix of slice lengths
    float *seg_v1 = (float *)(seg_pre + NT + 64);  // [NT] m1 value of the segment
    int *hist = (int *)(seg_v1 + NT);           // [256]
    int *wsum = hist + 256;                     // [64]
    int *sh = wsum + 64;                        // [SH_N .. 16]
    u64 *U = U_LDS ? (u64 *)(sh + 16) : (p.gU + (size_t)blockIdx.x * (size_t)p.cap);

    for (int i = tid; i < T; i += NT) { keys[i] = EMPTY; vals[i] = 0.f; }
    if (tid == 0) {
        sh[SH_CNT] = 0;
        sh[SH_OVF] = 0;
        sh[SH_NEXT] = p.static_sched ? (int)blockIdx.x : (int)atomicAdd(&p.queue[0], 1u);
    }
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    unsigned local_passes = 0;

    for (;;) {
        const int qi = sh[SH_NEXT];
        if (qi >= p.n_targets) break;
        const int slot_i = p.order ? p.order[qi] : qi;
        const int t = p.targets[slot_i];
        const int s1 = p.m1_indptr[t];
        const int n1 = p.m1_indptr[t + 1] - s1;

        // prefetch the next queue entry early; it is consumed at the bottom of the loop
        int next_q = 0;
        if (tid == 0) next_q = p.static_sched ? qi + (int)gridDim.x : (int)atomicAdd(&p.queue[0], 1u);

        Epi epi;
        epi.a1 = p.a1; epi.l1 = p.l1; epi.l2 = p.l2; epi.l3 = p.l3; epi.t1 = p.t1; epi.t2 = p.t2;
        epi.stab = p.stab; epi.bayes = p.bayes; epi.any = any_norm;
        epi.xtv = (p.l1 != 0.f) ? p.Xtv[t] : 0.f;
        epi.xcos = (p.l2 != 0.f) ? p.Xcos[t] : 0.f;
        epi.xdep = (p.l3 != 0.f) ? p.Xdep[t] : 0.f;
        epi.Ytv = p.Ytv; epi.Ycos = p.Ycos; epi.Ydep = p.Ydep;

        int f0 = 0, f1 = 0, g0 = 0, g1 = 0;
        if (p.filter_mode == SP_SEL_MATRIX) { f0 = p.f_indptr[t]; f1 = p.f_indptr[t + 1]; }
        if (p.target_mode == SP_SEL_MATRIX) { g0 = p.t_indptr[t]; g1 = p.t_indptr[t + 1]; }

        // ---- work estimate: MACs(t) = sum_u nnz(m2 row u) (upper bound on distinct candidates) ----
        u64 macs_local = 0;
        for (int j = tid; j < n1; j += NT) {
            const int u = p.m1_indices[s1 + j];
            macs_local += (u64)(p.m2_indptr[u + 1] - p.m2_indptr[u]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) macs_local += __shfl_xor(macs_local, d, 64);
        if (lane == 0) ((u64 *)wsum)[wave] = macs_local;  // wsum is 8-byte aligned, NW <= 16 -> 128 B
        __syncthreads();
        u64 macs = 0;
        for (int w = 0; w < NW; ++w) macs += ((u64 *)wsum)[w];
        __syncthreads();

        bool have_thr = false;
        unsigned thr_key = 0;
        int ub = 0;  // upper bound of the candidate count, see drain

        // ---- choose the column window width ----
        // dense windows can never overflow (one slot per column); hash windows are sized from the
        // MACs bound and split on overflow.  Window width w; windows are [lo, lo+w).
        long long width;
        if (p.n_cols <= T) {
            width = p.n_cols;
        } else {
            const long long p_dense = ((long long)p.n_cols + T - 1) / T;
            const long long p_hash = (long long)((macs + (u64)p.hash_fill - 1) / (u64)p.hash_fill);
            if (p_hash < 1 || p_dense <= p_hash) width = T;
            else width = ((long long)p.n_cols + p_hash - 1) / p_hash;
        }

        long long lo = 0;
        if (macs == 0) lo = p.n_cols;  // nothing to accumulate: empty output row

        while (lo < (long long)p.n_cols) {
            long long hi = lo + width;
            if (hi > p.n_cols) hi = p.n_cols;
            const int wlo = (int)lo, whi = (int)hi;
            const bool dense = (hi - lo) <= (long long)T;
            const bool whole = (wlo == 0 && whi == p.n_cols);
            int t_eff = dense ? (whi - wlo) : T;
            int hshift = 32 - p.logT;
            if (!dense && whole) {
                // single hash window over a small row: shrink the table so the drain scans less
                int lg = 10;
                while (lg < p.logT && (1ull << lg) < 2ull * macs) ++lg;
                t_eff = 1 << lg;
                hshift = 32 - lg;
            }
            const unsigned hmask = (unsigned)t_eff - 1u;

            // ================= accumulate =================
            for (int b0 = 0; b0 < n1; b0 += NT) {
                const int nb = min(NT, n1 - b0);
                int len = 0;
                if (tid < nb) {
                    const int u = p.m1_indices[s1 + b0 + tid];
                    int r0 = p.m2_indptr[u], r1 = p.m2_indptr[u + 1];
                    if (!whole && r0 < r1) {
                        // slice of the sorted m2 row inside [wlo, whi)  (s_plus.h:385-394)
                        r0 = lower_bound_g(p.m2_indices, r0, r1, wlo);
                        r1 = lower_bound_g(p.m2_indices, r0, r1, whi);
                    }
                    seg_lo[tid] = r0;
                    seg_v1[tid] = p.m1_data[s1 + b0 + tid];
                    len = r1 - r0;
                }
                const int incl = wave_incl_scan(len);
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                int woff = 0, total = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const int s = wsum[w];
                    if (w < wave) woff += s;
                    total += s;
                }
                seg_pre[tid] = woff + incl - len;
                if (tid == 0) seg_pre[NT] = total;
                __syncthreads();

                // flat element space [0,total): wave w owns a contiguous, 64-aligned chunk
                const int chunk = ((total + NW * 64 - 1) / (NW * 64)) * 64;
                const int e0 = wave * chunk;
                const int e1 = min(e0 + chunk, total);
                if (e0 < e1) {
                    // segment of this lane's first element: last s in [0,nb) with seg_pre[s] <= e
                    int e = e0 + lane;
                    int sl = 0, sr = nb;  // invariant: seg_pre[sl] <= e (seg_pre[0] = 0)
                    while (sr - sl > 1) {
                        const int mid = (sl + sr) >> 1;
                        if (seg_pre[mid] <= e) sl = mid; else sr = mid;
                    }
                    int seg = sl;
                    for (; e < e1; e += 64) {
                        while (seg + 1 < nb && e >= seg_pre[seg + 1]) ++seg;
                        const int j = seg_lo[seg] + (e - seg_pre[seg]);
                        const int c = p.m2_indices[j];
                        const float x = p.m2_data[j] * seg_v1[seg];
                        if (dense) {
                            const int s = c - wlo;
                            keys[s] = c;  // "touched" mark; all writers store the same value
                            atomicAdd(&vals[s], x);
                        } else {
                            unsigned s = ((unsigned)c * 2654435761u) >> hshift;
                            int probe = 0;
                            for (; probe < MAX_PROBE; ++probe) {
                                int cur = ((volatile int *)keys)[s];
                                if (cur == EMPTY) {
                                    const int prev = atomicCAS(&keys[s], EMPTY, c);
                                    cur = (prev == EMPTY) ? c : prev;
                                }
                                if (cur == c) { atomicAdd(&vals[s], x); break; }
                                s = (s + 1) & hmask;
                            }
                            if (probe == MAX_PROBE) sh[SH_OVF] = 1;
                        }
                    }
                }
                __syncthreads();  // seg_* are rewritten by the next batch
            }

            // ================= overflow: discard the window, halve it, retry =================
            if (!dense) {
                const int ovf = sh[SH_OVF];
                __syncthreads();
                if (ovf) {
                    for (int i = tid; i < t_eff; i += NT) { keys[i] = EMPTY; vals[i] = 0.f; }
                    if (tid == 0) sh[SH_OVF] = 0;
                    width = max((long long)T, (width + 1) / 2);
                    __syncthreads();
                    continue;
                }
            }
            ++local_passes;

            // ================= drain: selectors, epilogue, threshold, running top-k =================
            for (int base = 0; base < t_eff; base += NT * DRAIN_UNROLL) {
                // `ub` is a register-resident (hence workgroup-uniform) upper bound of SH_CNT; the exact
                // count is only consulted, between two barriers, when the bound says the buffer may fill.
                if (ub + NT * DRAIN_UNROLL > p.cap) {
                    __syncthreads();
                    const int n_now = sh[SH_CNT];
                    __syncthreads();
                    if (n_now + NT * DRAIN_UNROLL > p.cap) {
                        compact_topk<NT>(U, hist, sh, p.k, have_thr, thr_key);
                        ub = min(n_now, p.k);
                    } else {
                        ub = n_now;
                    }
                }
                ub += NT * DRAIN_UNROLL;
#pragma unroll
                for (int un = 0; un < DRAIN_UNROLL; ++un) {
                    const int s = base + un * NT + tid;
                    bool want = false;
                    u64 item = 0;
                    if (s < t_eff) {
                        const int c = keys[s];
                        if (c != EMPTY) {
                            const float xy = vals[s];
                            keys[s] = EMPTY;
                            vals[s] = 0.f;
                            bool pass = true;
                            if (p.filter_mode == SP_SEL_MATRIX) pass = !range_has(p.f_indices, f0, f1, c);
                            if (pass && p.target_mode == SP_SEL_MATRIX) pass = range_has(p.t_indices, g0, g1, c);
                            if (pass) {
                                const float val = epi(c, xy);
                                if (val >= p.threshold) {
                                    const unsigned key = fkey(val);
                                    if (!have_thr || key > thr_key) {
                                        want = true;
                                        item = ((u64)key << 32) | (u64)(unsigned)c;
                                    }
                                }
                            }
                        }
                    }
                    const u64 m = __ballot(want);
                    if (m) {
                        int wbase = 0;
                        if (lane == 0) wbase = atomicAdd(&sh[SH_CNT], __popcll(m));
                        wbase = __shfl(wbase, 0, 64);
                        if (want) U[wbase + __popcll(m & ((1ull << lane) - 1ull))] = item;
                    }
                }
            }
            lo = hi;
        }

        // ================= final selection + write-out =================
        __syncthreads();
        if (sh[SH_CNT] > p.k) compact_topk<NT>(U, hist, sh, p.k, have_thr, thr_key);
        const int n_out = sh[SH_CNT];
        const long long o = (long long)slot_i * (long long)p.k;
        for (int j = tid; j < p.k; j += NT) {
            int r = 0, c = 0;
            float v = 0.f;
            if (j < n_out) {
                const u64 it = U[j];
                r = t;
                c = (int)(unsigned)(it & 0xFFFFFFFFull);
                v = funkey((unsigned)(it >> 32));
            }
            if (p.rows) p.rows[o + j] = r;
            p.cols[o + j] = c;
            p.values[o + j] = v;
        }
        if (tid == 0) {
            if (p.counts) p.counts[slot_i] = n_out;
            sh[SH_NEXT] = next_q;
        }
        __syncthreads();
        if (tid == 0) sh[SH_CNT] = 0;
        __syncthreads();
    }
    if (p.count_passes && tid == 0 && local_passes) atomicAdd(&p.queue[1], local_passes);
}

// Work-sorted visiting order: key = MACs(t) clamped to 32 bits, computed per slot.
__global__ void sp_row_work_kernel(int n_targets, const int *targets, const int *m1_indices, const int *m1_indptr,
                                   const int *m2_indptr, unsigned *work) {
    const int gw = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (gw >= n_targets) return;
    const int t = targets[gw];
    const int s = m1_indptr[t], e = m1_indptr[t + 1];
    u64 acc = 0;
    for (int j = s + lane; j < e; j += 64) {
        const int u = m1_indices[j];
        acc += (u64)(m2_indptr[u + 1] - m2_indptr[u]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) work[gw] = acc > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)acc;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side: C ABI
// ---------------------------------------------------------------------------------------------
namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(SP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct Config {
    int T, logT, NT, num_wgs, cap, hash_fill;
    bool u_lds;
    size_t lds_bytes;
    size_t ws_gu_bytes;     // candidate buffers in global memory (0 when in LDS)
    size_t ws_total;        // queue + gU (+ order/work when sorted scheduling is on)
};

constexpr size_t WS_QUEUE_BYTES = 256;
constexpr size_t LDS_LIMIT = 160 * 1024;

size_t lds_fixed_bytes(int T, int NT) {
    // keys + vals + seg_lo + seg_pre(+64) + seg_v1 + hist + wsum + sh
    return (size_t)T * 8 + (size_t)NT * 4 + (size_t)(NT + 64) * 4 + (size_t)NT * 4 + 256 * 4 + 64 * 4 + 16 * 4;
}

int make_config(const sp_knn_args *a, int n_cus, Config *c) {
    int NT = a->threads_per_wg ? a->threads_per_wg : 512;
    if (NT != 256 && NT != 512 && NT != 1024) return fail(SP_EINVAL, "threads_per_wg must be 256, 512 or 1024 (got %d)", NT);
    int T = a->table_slots ? a->table_slots : 16384;
    if (T < 1024 || (T & (T - 1))) return fail(SP_EINVAL, "table_slots must be a power of two >= 1024 (got %d)", T);
    int logT = 0;
    while ((1 << logT) < T) ++logT;
    const int load = a->load_pct > 0 ? std::min(a->load_pct, 90) : 50;

    const long long need_cap = (long long)a->k + (long long)NT * DRAIN_UNROLL;
    size_t fixed = lds_fixed_bytes(T, NT);
    if (fixed + 8 * 1024 > LDS_LIMIT) return fail(SP_EINVAL, "table_slots=%d does not fit the 160 KiB LDS", T);
    // candidate buffer: LDS if (k + NT*UNROLL) entries fit beside the table, else global scratch
    long long cap_lds = (long long)((LDS_LIMIT - fixed) / 8);
    bool u_lds = need_cap <= cap_lds;
    long long cap;
    if (u_lds) {
        cap = std::max<long long>(need_cap, std::min<long long>(cap_lds, 2048));
    } else {
        cap = need_cap + 1024;
    }
    if (cap > 0x7FFFFFF0LL) return fail(SP_EINVAL, "k too large");
    c->T = T; c->logT = logT; c->NT = NT; c->cap = (int)cap; c->u_lds = u_lds;
    c->hash_fill = std::max(1, (int)((long long)T * load / 100));
    c->lds_bytes = fixed + (u_lds ? (size_t)cap * 8 : 0);
    int wgs_per_cu = (int)std::max<size_t>(1, LDS_LIMIT / c->lds_bytes);
    wgs_per_cu = std::min(wgs_per_cu, 2048 / NT);
    wgs_per_cu = std::max(1, std::min(wgs_per_cu, 8));
    int num_wgs = a->num_wgs > 0 ? a->num_wgs : n_cus * wgs_per_cu;
    num_wgs = std::max(1, std::min(num_wgs, std::max(1, a->n_targets)));
    c->num_wgs = num_wgs;
    c->ws_gu_bytes = u_lds ? 0 : (size_t)num_wgs * (size_t)cap * 8;
    c->ws_total = WS_QUEUE_BYTES + c->ws_gu_bytes;
    return SP_OK;
}

int validate(const sp_knn_args *a) {
    if (!a) return fail(SP_EINVAL, "args is NULL");
    if (a->struct_size != sizeof(sp_knn_args))
        return fail(SP_EINVAL, "sp_knn_args size mismatch: caller %u, library %zu", a->struct_size, sizeof(sp_knn_args));
    if (a->n_targets < 0 || a->n_rows_m1 < 0 || a->n_rows_m2 < 0 || a->n_output_cols < 0)
        return fail(SP_EINVAL, "negative dimension");
    if (a->k < 1) return fail(SP_EINVAL, "k must be >= 1, got %d", a->k);
    if (a->nnz_m1 < 0 || a->nnz_m2 < 0 || a->nnz_m1 > 0x7FFFFFFFLL || a->nnz_m2 > 0x7FFFFFFFLL)
        return fail(SP_EINVAL, "nnz must fit int32 indptr (reference limit, s_plus.pyx:241-244)");
    if (a->n_targets > 0) {
        if (!a->targets || !a->m1_indptr || !a->m2_indptr || !a->cols || !a->values)
            return fail(SP_EINVAL, "NULL input/output pointer");
        if (!a->rows && !(a->on_device && (a->flags & SP_FLAG_NO_ROWS_OUT)))
            return fail(SP_EINVAL, "rows is NULL");
        if (a->nnz_m1 > 0 && (!a->m1_data || !a->m1_indices)) return fail(SP_EINVAL, "m1 arrays NULL");
        if (a->nnz_m2 > 0 && (!a->m2_data || !a->m2_indices)) return fail(SP_EINVAL, "m2 arrays NULL");
        if (a->l1 != 0.f && (!a->Xtversky || !a->Ytversky)) return fail(SP_EINVAL, "l1 != 0 needs Xtversky/Ytversky");
        if (a->l2 != 0.f && (!a->Xcosine || !a->Ycosine)) return fail(SP_EINVAL, "l2 != 0 needs Xcosine/Ycosine");
        if (a->l3 != 0.f && (!a->Xdepop || !a->Ydepop)) return fail(SP_EINVAL, "l3 != 0 needs Xdepop/Ydepop");
        if (a->filter_mode == SP_SEL_MATRIX && (!a->filter_m_indptr || (a->filter_nnz > 0 && !a->filter_m_indices)))
            return fail(SP_EINVAL, "filter MATRIX mode needs indptr/indices");
        if (a->target_col_mode == SP_SEL_MATRIX && (!a->target_col_m_indptr || (a->target_col_nnz > 0 && !a->target_col_m_indices)))
            return fail(SP_EINVAL, "target MATRIX mode needs indptr/indices");
    }
    if (a->filter_mode < 0 || a->filter_mode > 2 || a->target_col_mode < 0 || a->target_col_mode > 2)
        return fail(SP_EINVAL, "bad selector mode");
    return SP_OK;
}

int device_cus(int device, int *n_cus) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    *n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return SP_OK;
}

template <int NT, bool U_LDS>
int launch_rows(const KParams &kp, const Config &c, hipStream_t stream) {
    auto kern = sp_knn_rows_kernel<NT, U_LDS>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_bytes));
    hipLaunchKernelGGL(kern, dim3(c.num_wgs), dim3(NT), c.lds_bytes, stream, kp);
    HIP_TRY(hipGetLastError());
    return SP_OK;
}

// all pointers in `a` are device pointers here
int run_device(sp_knn_args *a) {
    HIP_TRY(hipSetDevice(a->device));
    if (a->n_targets == 0) { a->kernel_ms = 0.f; return SP_OK; }
    int n_cus = 256;
    int rc = device_cus(a->device, &n_cus);
    if (rc) return rc;
    Config c;
    rc = make_config(a, n_cus, &c);
    if (rc) return rc;

    hipStream_t stream = (hipStream_t)a->stream;
    unsigned char *ws = (unsigned char *)a->workspace;
    bool own_ws = false;
    if (!ws) {
        HIP_TRY(hipMalloc((void **)&ws, c.ws_total));
        own_ws = true;
    } else if (a->workspace_bytes < (int64_t)c.ws_total) {
        return fail(SP_EWORKSPACE, "workspace too small: need %zu bytes, got %lld", c.ws_total, (long long)a->workspace_bytes);
    }

    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }

    HIP_TRY(hipMemsetAsync(ws, 0, WS_QUEUE_BYTES, stream));

    KParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.n_targets = a->n_targets; kp.targets = a->targets;
    kp.m1_data = a->m1_data; kp.m1_indices = a->m1_indices; kp.m1_indptr = a->m1_indptr;
    kp.m2_data = a->m2_data; kp.m2_indices = a->m2_indices; kp.m2_indptr = a->m2_indptr;
    kp.Xtv = a->Xtversky; kp.Ytv = a->Ytversky; kp.Xcos = a->Xcosine; kp.Ycos = a->Ycosine;
    kp.Xdep = a->Xdepop; kp.Ydep = a->Ydepop;
    kp.a1 = a->a1; kp.l1 = a->l1; kp.l2 = a->l2; kp.l3 = a->l3; kp.t1 = a->t1; kp.t2 = a->t2;
    kp.stab = a->stabilized_shrink; kp.bayes = a->bayesian_shrink; kp.threshold = a->threshold;
    kp.k = a->k; kp.n_cols = a->n_output_cols;
    kp.filter_mode = a->filter_mode; kp.f_indptr = a->filter_m_indptr; kp.f_indices = a->filter_m_indices;
    kp.target_mode = a->target_col_mode; kp.t_indptr = a->target_col_m_indptr; kp.t_indices = a->target_col_m_indices;
    kp.rows = a->rows; kp.cols = a->cols; kp.values = a->values; kp.counts = a->out_counts;
    kp.T = c.T; kp.logT = c.logT; kp.cap = c.cap;
    kp.queue = (unsigned int *)ws;
    kp.gU = c.u_lds ? nullptr : (u64 *)(ws + WS_QUEUE_BYTES);
    kp.order = nullptr;
    kp.hash_fill = c.hash_fill;
    kp.static_sched = (a->flags & SP_FLAG_STATIC_SCHED) ? 1 : 0;
    kp.count_passes = timed ? 1 : 0;

    if (c.NT == 256) rc = c.u_lds ? launch_rows<256, true>(kp, c, stream) : launch_rows<256, false>(kp, c, stream);
    else if (c.NT == 512) rc = c.u_lds ? launch_rows<512, true>(kp, c, stream) : launch_rows<512, false>(kp, c, stream);
    else rc = c.u_lds ? launch_rows<1024, true>(kp, c, stream) : launch_rows<1024, false>(kp, c, stream);
    if (rc) { if (own_ws) (void)hipFree(ws); return rc; }

    if (timed) {
        HIP_TRY(hipEventRecord(ev1, stream));
        HIP_TRY(hipEventSynchronize(ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        a->kernel_ms = ms;
        unsigned q[2] = {0, 0};
        HIP_TRY(hipMemcpy(q, ws, sizeof(q), hipMemcpyDeviceToHost));
        a->passes_total = (int32_t)q[1];
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
    }
    if (own_ws) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipFree(ws));
    }
    return SP_OK;
}

// RAII device allocation list for the host-pointer entry
struct DevPool {
    std::vector<void *> ptrs;
    ~DevPool() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename Tp>
    int up(const Tp *host, size_t n, const Tp **dev) {
        *dev = nullptr;
        if (!host || n == 0) {
            // keep a valid (1-element) device pointer so kernels never see host addresses
            void *d = nullptr;
            HIP_TRY(hipMalloc(&d, sizeof(Tp)));
            ptrs.push_back(d);
            *dev = (const Tp *)d;
            return SP_OK;
        }
        void *d = nullptr;
        HIP_TRY(hipMalloc(&d, n * sizeof(Tp)));
        ptrs.push_back(d);
        HIP_TRY(hipMemcpy(d, host, n * sizeof(Tp), hipMemcpyHostToDevice));
        *dev = (const Tp *)d;
        return SP_OK;
    }
    template <typename Tp>
    int alloc(size_t n, Tp **dev) {
        void *d = nullptr;
        HIP_TRY(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(Tp)));
        ptrs.push_back(d);
        *dev = (Tp *)d;
        return SP_OK;
    }
};

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// host pointers in, host pointers out: the drop-in for s_plus.pyx:359-384
int run_host(sp_knn_args *a) {
    HIP_TRY(hipSetDevice(a->device));
    const size_t nt = (size_t)a->n_targets, k = (size_t)a->k;
    if (nt == 0) return SP_OK;
    // the reference trusts `targets` (s_plus.pyx:191-196, no bounds check); a device kernel must not
    for (size_t i = 0; i < nt; ++i)
        if (a->targets[i] < 0 || a->targets[i] >= a->n_rows_m1)
            return fail(SP_EINVAL, "targets[%zu]=%d out of range [0,%d)", i, a->targets[i], a->n_rows_m1);

    DevPool pool;
    sp_knn_args d = *a;
    d.on_device = 1;
    d.stream = nullptr;
    d.workspace = nullptr;
    d.workspace_bytes = 0;
    TRY(pool.up(a->targets, nt, &d.targets));
    TRY(pool.up(a->m1_data, (size_t)a->nnz_m1, &d.m1_data));
    TRY(pool.up(a->m1_indices, (size_t)a->nnz_m1, &d.m1_indices));
    TRY(pool.up(a->m1_indptr, (size_t)a->n_rows_m1 + 1, &d.m1_indptr));
    TRY(pool.up(a->m2_data, (size_t)a->nnz_m2, &d.m2_data));
    TRY(pool.up(a->m2_indices, (size_t)a->nnz_m2, &d.m2_indices));
    TRY(pool.up(a->m2_indptr, (size_t)a->n_rows_m2 + 1, &d.m2_indptr));
    TRY(pool.up(a->l1 != 0.f ? a->Xtversky : nullptr, (size_t)a->n_rows_m1, &d.Xtversky));
    TRY(pool.up(a->l1 != 0.f ? a->Ytversky : nullptr, (size_t)a->n_output_cols, &d.Ytversky));
    TRY(pool.up(a->l2 != 0.f ? a->Xcosine : nullptr, (size_t)a->n_rows_m1, &d.Xcosine));
    TRY(pool.up(a->l2 != 0.f ? a->Ycosine : nullptr, (size_t)a->n_output_cols, &d.Ycosine));
    TRY(pool.up(a->l3 != 0.f ? a->Xdepop : nullptr, (size_t)a->n_rows_m1, &d.Xdepop));
    TRY(pool.up(a->l3 != 0.f ? a->Ydepop : nullptr, (size_t)a->n_output_cols, &d.Ydepop));
    const bool fm = a->filter_mode == SP_SEL_MATRIX, tm = a->target_col_mode == SP_SEL_MATRIX;
    TRY(pool.up(fm ? a->filter_m_indptr : nullptr, (size_t)a->n_rows_m1 + 1, &d.filter_m_indptr));
    TRY(pool.up(fm ? a->filter_m_indices : nullptr, (size_t)a->filter_nnz, &d.filter_m_indices));
    TRY(pool.up(tm ? a->target_col_m_indptr : nullptr, (size_t)a->n_rows_m1 + 1, &d.target_col_m_indptr));
    TRY(pool.up(tm ? a->target_col_m_indices : nullptr, (size_t)a->target_col_nnz, &d.target_col_m_indices));
    TRY(pool.alloc(nt * k, &d.rows));
    TRY(pool.alloc(nt * k, &d.cols));
    TRY(pool.alloc(nt * k, &d.values));
    d.out_counts = nullptr;
    if (a->out_counts) TRY(pool.alloc(nt, &d.out_counts));

    int rc = run_device(&d);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(a->rows, d.rows, nt * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(a->cols, d.cols, nt * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(a->values, d.values, nt * k * sizeof(float), hipMemcpyDeviceToHost));
    if (a->out_counts) HIP_TRY(hipMemcpy(a->out_counts, d.out_counts, nt * sizeof(int32_t), hipMemcpyDeviceToHost));
    a->kernel_ms = d.kernel_ms;
    a->passes_total = d.passes_total;
    return SP_OK;
}

}  // namespace

extern "C" {

int sp_abi_version(void) { return SP_KNN_ABI_VERSION; }

const char *sp_last_error(void) { return g_err; }

int sp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int sp_backend_info(int device, char *buf, int buflen) {
    if (!buf || buflen <= 0) return fail(SP_EINVAL, "bad buffer");
    if (sp_device_count() <= 0) return fail(SP_ENODEVICE, "no HIP device");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    int n = snprintf(buf, (size_t)buflen, "%s arch=%s CUs=%d LDS/WG=%zu HBM=%.1fGiB clock=%dMHz", prop.name, prop.gcnArchName,
                     prop.multiProcessorCount, (size_t)prop.sharedMemPerBlock, (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0),
                     prop.clockRate / 1000);
    return n;
}

int64_t sp_knn_workspace_bytes(const sp_knn_args *a) {
    int rc = validate(a);
    if (rc) return rc;
    int n_cus = 256;
    if (sp_device_count() > 0) {
        rc = device_cus(a->device, &n_cus);
        if (rc) return rc;
    }
    Config c;
    rc = make_config(a, n_cus, &c);
    if (rc) return rc;
    return (int64_t)c.ws_total;
}

int sp_knn_f32_i32(sp_knn_args *a) {
    g_err[0] = 0;
    int rc = validate(a);
    if (rc) return rc;
    const int ndev = sp_device_count();
    if (ndev <= 0) return fail(SP_ENODEVICE, "no HIP device visible: similaripy_amd has no CPU fallback");
    if (a->device < 0 || a->device >= ndev) return fail(SP_EINVAL, "device %d out of range (have %d)", a->device, ndev);
    return a->on_device ? run_device(a) : run_host(a);
}

}  // extern "C"
