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
//     LDS-resident accumulator tile of T 64-bit {column, partial sum} slots updated with
//     ds_cmpst_rtn_b64 only (measured on gfx950: ds_add_f32 retires 0.33 lanes/clk/CU whatever the
//     address pattern, compare-and-swap 3-6 lanes/clk — scripts/lds_atomics_bench.hip):
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
constexpr u64 EMPTY64 = 0xFFFFFFFF00000000ull;   // free slot: key EMPTY, partial sum +0.0f
constexpr u64 NOSLOT64 = 0xFFFFFFFF7FC0DEADull;  // never stored (EMPTY key with a non-zero sum): a CAS expecting it is a no-op
// table slots per thread per drain iteration (their Y gathers fly together); 1024-thread workgroups have
// half the VGPR budget (128), where 8 would spill
#define DRAIN_UNROLL (NT >= 1024 ? 4 : 8)
constexpr int ACC_UNROLL = 4;    // m2 elements per lane kept in flight in the accumulate loop
constexpr int U_SLACK = 1024;    // candidate-buffer entries beyond k (room between two selections)
constexpr int MAX_PROBE = 128;   // linear-probe budget before a window is declared overflowed

// scalar slots in LDS
enum { SH_CNT = 0, SH_OVF, SH_SEL, SH_NEED, SH_EQ, SH_CNT2, SH_NEXT, SH_RETRY, SH_N };

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
    int cap;               // candidate buffer capacity (> k)
    u64 *gU;               // candidate buffers in global memory (only when they do not fit LDS)
    unsigned int *queue;   // [0] = next slot index (dynamic scheduling), [1] = pass counter (debug)
    const int *order;      // optional: slot visiting order (descending work); NULL = identity
    int hash_fill;         // slots' worth of MACs one hash window may receive (= T * load_pct / 100)
    int static_sched;
    int count_passes;
    unsigned long long *phase_cycles;  // optional [PH_N]: s_memtime cycles of workgroup lane 0 per phase
    int dbg;               // ablation bits for profiling only (results are WRONG when non-zero):
                           // 1 = accumulate: no LDS inserts, 2 = accumulate: no global loads, 4 = drain: no Y gathers
};

// phases timed by lane 0 of every workgroup when KParams::phase_cycles != NULL
enum { PH_SETUP = 0, PH_SEGMENTS, PH_ACCUM, PH_DRAIN, PH_SELECT, PH_OUTPUT, PH_N,
       CT_ROUNDS = PH_N, CT_ITERS, CT_OVF, CT_SWEEPS, CT_N };

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
    // ytv / ycos / ydep: the column terms Ytv[col] / Ycos[col] / Ydep[col], gathered by the caller so
    // that the loads of several candidates are in flight together (0 where the weight is 0)
    __device__ __forceinline__ float operator()(float xy, float ytv, float ycos, float ydep) const {
        float vt = 0.f, vc = 0.f, vd = 0.f, val = xy;
        if (l1 != 0.f) vt = l1 * (t1 * (xtv - xy) + t2 * (ytv - xy) + xy);
        if (l2 != 0.f) vc = l2 * (xcos * ycos);
        if (l3 != 0.f) vd = l3 * (xdep * ydep);
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
    u64 *tab = (u64 *)smem;                     // [T]  {column id : partial dot product}
    int *seg_lo = (int *)(tab + T);             // [NT]   start of the window's slice of m2 row u
    int *seg_pre = seg_lo + NT;                 // [NT+64] exclusive prefix of slice lengths
    float *seg_v1 = (float *)(seg_pre + NT + 64);  // [NT] m1 value of the segment
    int *seg_hi = (int *)(seg_v1 + NT);         // [NT]   end of the slice (= start of the next window's)
    int *hist = seg_hi + NT;                    // [256]
    int *wsum = hist + 256;                     // [64]
    int *sh = wsum + 64;                        // [SH_N .. 16]
    u64 *U = U_LDS ? (u64 *)(sh + 16) : (p.gU + (size_t)blockIdx.x * (size_t)p.cap);

    for (int i = tid; i < T; i += NT) tab[i] = EMPTY64;
    if (tid == 0) {
        sh[SH_CNT] = 0;
        sh[SH_OVF] = 0;
        sh[SH_RETRY] = 0;
        sh[SH_NEXT] = p.static_sched ? (int)blockIdx.x : (int)atomicAdd(&p.queue[0], 1u);
    }
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    unsigned local_passes = 0;
    // phase timers (lane 0 only; s_memtime ticks are shader cycles)
    const bool timing = (p.phase_cycles != nullptr) && tid == 0;
    u64 ph[CT_N] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 tmark = timing ? (u64)clock64() : 0;
#define PHASE_END(which) do { if (timing) { const u64 _n = (u64)clock64(); ph[which] += _n - tmark; tmark = _n; } } while (0)

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
        PHASE_END(PH_SETUP);

        bool have_thr = false;
        unsigned thr_key = 0;
        bool retry_window = false;  // the current window repeats the previous lo (after an overflow)

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
            // Window slices chain (hi of window w == lo of window w+1), so when the m1 row fits one
            // batch the previous slice end is kept in LDS and only one lower_bound per window is run.
            const bool carry = (n1 <= NT);
            for (int b0 = 0; b0 < n1; b0 += NT) {
                const int nb = min(NT, n1 - b0);
                int len = 0;
                if (tid < nb) {
                    const int u = p.m1_indices[s1 + b0 + tid];
                    int r0 = p.m2_indptr[u], r1 = p.m2_indptr[u + 1];
                    if (!whole) {
                        // slice of the sorted m2 row inside [wlo, whi)  (s_plus.h:385-394)
                        if (wlo != 0) {
                            if (carry) r0 = retry_window ? seg_lo[tid] : seg_hi[tid];
                            else r0 = lower_bound_g(p.m2_indices, r0, r1, wlo);
                        }
                        if (whi < p.n_cols) r1 = lower_bound_g(p.m2_indices, r0, r1, whi);
                        if (carry) seg_hi[tid] = r1;
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
                    const int sw = wsum[w];
                    if (w < wave) woff += sw;
                    total += sw;
                }
                seg_pre[tid] = woff + incl - len;
                if (tid == 0) seg_pre[NT] = total;
                __syncthreads();
                PHASE_END(PH_SEGMENTS);

                // flat element space [0,total): wave w owns a contiguous, 64-aligned chunk; every lane
                // keeps ACC_UNROLL coalesced (stride-64) index/value loads in flight
                const int chunk = ((total + NW * 64 - 1) / (NW * 64)) * 64;
                const int e0 = wave * chunk;
                const int e1 = min(e0 + chunk, total);
                if (e0 < e1) {
                    int e = e0 + lane;
                    int sl = 0, sr = nb;  // last s in [0,nb) with seg_pre[s] <= e (seg_pre[0] = 0)
                    while (sr - sl > 1) {
                        const int mid = (sl + sr) >> 1;
                        if (seg_pre[mid] <= e) sl = mid; else sr = mid;
                    }
                    int seg = sl;
                    for (; e < e1; e += 64 * ACC_UNROLL) {
                        int idx[ACC_UNROLL];
                        float v1[ACC_UNROLL];
#pragma unroll
                        for (int j = 0; j < ACC_UNROLL; ++j) {
                            const int ej = e + 64 * j;
                            if (ej < e1) {
                                while (seg + 1 < nb && ej >= seg_pre[seg + 1]) ++seg;
                                idx[j] = seg_lo[seg] + (ej - seg_pre[seg]);
                                v1[j] = seg_v1[seg];
                            } else {
                                // padding: repeat the lane's first element with weight 0 — it goes through
                                // the same (branch-free) insert path and adds exactly 0.0 to an existing key
                                idx[j] = idx[0];
                                v1[j] = 0.f;
                            }
                        }
                        int c[ACC_UNROLL];
                        float x[ACC_UNROLL];
                        if (p.dbg & 2) {
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) { c[j] = wlo + (int)(((unsigned)idx[j] * 40503u) % (unsigned)(whi - wlo)); x[j] = 1.f; }
                        } else {
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) { c[j] = p.m2_indices[idx[j]]; x[j] = p.m2_data[idx[j]]; }
                        }
#pragma unroll
                        for (int j = 0; j < ACC_UNROLL; ++j) x[j] *= v1[j];
                        if (p.dbg & 1) {
                            float sink = 0.f;
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) sink += x[j] + (float)c[j];
                            if (sink == 123.456f) sh[SH_OVF] = 2;  // keeps the loads alive, never true in practice
                        } else if (dense) {
                            // direct-indexed window: every column owns its slot, no claim needed — mark the key
                            // half (all writers store the same value) and add into the sum half
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) {
                                unsigned *slot32 = (unsigned *)&tab[c[j] - wlo];
                                slot32[1] = (unsigned)c[j];
                                atomicAdd((float *)slot32, x[j]);
                            }
                        } else {
                            // Hashed window.  One 64-bit compare-and-swap claims a free slot for a new column AND
                            // deposits its first product; finding the same column already there turns into a
                            // hardware float add on the sum half (slow on gfx950, 3 clk/lane, but immune to
                            // contention on hot columns); finding another column means linear probing.
                            // Round 1 issues the ACC_UNROLL claims back to back; the few leftovers are then walked
                            // one element per lane per round, so later rounds are sparse instructions.
                            unsigned hs[ACC_UNROLL];
                            u64 prev[ACC_UNROLL];
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) hs[j] = ((unsigned)c[j] * 2654435761u) >> hshift;
                            if (p.dbg & 32) {   // ablation: collision-free slots (every claim succeeds)
#pragma unroll
                                for (int j = 0; j < ACC_UNROLL; ++j) hs[j] = (unsigned)(e + 64 * j) & hmask;
                            }
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j)
                                prev[j] = atomicCAS(&tab[hs[j]], EMPTY64, ((u64)(unsigned)c[j] << 32) | (u64)__float_as_uint(x[j]));
                            unsigned pend = 0;
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) {
                                const bool hit = ((int)(prev[j] >> 32) == c[j]);
                                if (hit && !(p.dbg & 16)) atomicAdd((float *)&tab[hs[j]], x[j]);
                                if (prev[j] != EMPTY64 && !hit) pend |= 1u << j;
                            }
                            if (p.dbg & 8) pend = 0;   // ablation: no probing beyond the home slot
                            int probes = 1, plen = 0;
                            while (__ballot(pend != 0)) {   // wave-uniform trip count
                                ++probes;
                                if (!pend) continue;
                                const unsigned bit = pend & (0u - pend);  // this lane's current element
                                int cc = c[0];
                                float xx = x[0];
                                unsigned hh = hs[0];
#pragma unroll
                                for (int j = 1; j < ACC_UNROLL; ++j)
                                    if (bit == (1u << j)) { cc = c[j]; xx = x[j]; hh = hs[j]; }
                                // double hashing: an odd, key-dependent stride visits every slot of the
                                // power-of-two table and avoids the long clusters of linear probing
                                hh = (hh + ((((unsigned)cc * 0x85EBCA6Bu) >> 15) | 1u)) & hmask;
                                const u64 pv = atomicCAS(&tab[hh], EMPTY64, ((u64)(unsigned)cc << 32) | (u64)__float_as_uint(xx));
                                const bool hit = ((int)(pv >> 32) == cc);
                                if (hit) atomicAdd((float *)&tab[hh], xx);
                                if (pv == EMPTY64 || hit) { pend &= ~bit; plen = 0; }
                                else if (++plen >= MAX_PROBE) { sh[SH_OVF] = 1; break; }
#pragma unroll
                                for (int j = 0; j < ACC_UNROLL; ++j)
                                    if (bit == (1u << j)) hs[j] = hh;
                            }
                            if (timing) { ph[CT_ROUNDS] += probes; ph[CT_ITERS] += 1; }
                        }
                    }
                }
                __syncthreads();  // seg_* are rewritten by the next batch
                PHASE_END(PH_ACCUM);
            }

            // ================= overflow: discard the window, halve it, retry =================
            if (!dense) {
                const int ovf = sh[SH_OVF];
                __syncthreads();
                if (ovf) {
                    if (timing) ph[CT_OVF] += 1;
                    for (int i = tid; i < t_eff; i += NT) tab[i] = EMPTY64;
                    if (tid == 0) sh[SH_OVF] = 0;
                    width = max((long long)T, (width + 1) / 2);
                    retry_window = true;  // same lo again: slice starts are still in seg_lo
                    __syncthreads();
                    continue;
                }
            }
            retry_window = false;
            ++local_passes;

            // ================= drain: selectors, epilogue, threshold, running top-k =================
            // One barrier-free sweep over the table.  Survivors of the running threshold are appended to
            // U; a survivor that finds U full leaves its slot in place and raises SH_RETRY, after which
            // the workgroup selects the k best of U (raising the threshold) and sweeps the leftovers.
            for (;;) {
                if (timing) ph[CT_SWEEPS] += 1;
                for (int base = 0; base < t_eff; base += NT * DRAIN_UNROLL) {
                    int c[DRAIN_UNROLL];
                    float xy[DRAIN_UNROLL];
                    unsigned occ = 0, pass = 0;
#pragma unroll
                    for (int j = 0; j < DRAIN_UNROLL; ++j) {
                        const int sidx = base + j * NT + tid;
                        c[j] = EMPTY;
                        xy[j] = 0.f;
                        if (sidx < t_eff) {
                            const u64 slot = tab[sidx];
                            c[j] = (int)(slot >> 32);
                            xy[j] = __uint_as_float((unsigned)slot);
                        }
                        if (c[j] != EMPTY) occ |= 1u << j;
                    }
                    pass = occ;
                    if (p.filter_mode == SP_SEL_MATRIX) {
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j)
                            if ((pass & (1u << j)) && range_has(p.f_indices, f0, f1, c[j])) pass &= ~(1u << j);
                    }
                    if (p.target_mode == SP_SEL_MATRIX) {
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j)
                            if ((pass & (1u << j)) && !range_has(p.t_indices, g0, g1, c[j])) pass &= ~(1u << j);
                    }
                    // gather the column terms of all DRAIN_UNROLL slots first (loads in flight together);
                    // slots that are empty / filtered read column 0 (one hot line)
                    float ytv[DRAIN_UNROLL], ycos[DRAIN_UNROLL], ydep[DRAIN_UNROLL];
                    int gc[DRAIN_UNROLL];
#pragma unroll
                    for (int j = 0; j < DRAIN_UNROLL; ++j) {
                        gc[j] = ((pass & (1u << j)) && !(p.dbg & 4)) ? c[j] : 0;
                        ytv[j] = 0.f; ycos[j] = 0.f; ydep[j] = 0.f;
                    }
                    if (p.l1 != 0.f) {
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) ytv[j] = p.Ytv[gc[j]];
                    }
                    if (p.l2 != 0.f) {
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) ycos[j] = p.Ycos[gc[j]];
                    }
                    if (p.l3 != 0.f) {
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) ydep[j] = p.Ydep[gc[j]];
                    }
                    float val[DRAIN_UNROLL];
#pragma unroll
                    for (int j = 0; j < DRAIN_UNROLL; ++j) val[j] = epi(xy[j], ytv[j], ycos[j], ydep[j]);
                    // survivors of (threshold, running k-th value) in this lane's DRAIN_UNROLL slots
                    unsigned want = 0;
                    unsigned key[DRAIN_UNROLL];
#pragma unroll
                    for (int j = 0; j < DRAIN_UNROLL; ++j) {
                        key[j] = fkey(val[j]);
                        if ((pass & (1u << j)) && (val[j] >= p.threshold) && (!have_thr || key[j] > thr_key)) want |= 1u << j;
                    }
                    // one aggregated reservation per wave: lane counts -> wave scan -> single LDS atomic
                    unsigned stored = 0;
                    if (__ballot(want != 0)) {
                        const int mine = __popc(want);
                        const int incl = wave_incl_scan(mine);
                        int wbase = 0;
                        if (lane == 63) wbase = atomicAdd(&sh[SH_CNT], incl);
                        wbase = __shfl(wbase, 63, 64);
                        int pos = wbase + incl - mine;
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) {
                            if (want & (1u << j)) {
                                if (pos < p.cap) {
                                    U[pos] = ((u64)key[j] << 32) | (u64)(unsigned)c[j];
                                    stored |= 1u << j;
                                } else {
                                    sh[SH_RETRY] = 1;  // U is full: keep the slot for the sweep after the selection
                                }
                                ++pos;
                            }
                        }
                    }
                    // free every visited slot except survivors that found U full
                    const unsigned clear = occ & (~want | stored);
#pragma unroll
                    for (int j = 0; j < DRAIN_UNROLL; ++j)
                        if (clear & (1u << j)) tab[base + j * NT + tid] = EMPTY64;
                }
                __syncthreads();  // sweep complete (also orders the slot clears before the next window)
                const int retry = sh[SH_RETRY];
                if (!retry) break;  // uniform
                __syncthreads();    // everyone has seen the flag
                if (tid == 0) {
                    sh[SH_RETRY] = 0;
                    if (sh[SH_CNT] > p.cap) sh[SH_CNT] = p.cap;  // failed appends over-counted
                }
                __syncthreads();
                PHASE_END(PH_DRAIN);
                compact_topk<NT>(U, hist, sh, p.k, have_thr, thr_key);
                PHASE_END(PH_SELECT);
            }
            PHASE_END(PH_DRAIN);
            lo = hi;
        }

        // ================= final selection + write-out =================
        __syncthreads();
        PHASE_END(PH_DRAIN);
        if (sh[SH_CNT] > p.k) compact_topk<NT>(U, hist, sh, p.k, have_thr, thr_key);
        PHASE_END(PH_SELECT);
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
        PHASE_END(PH_OUTPUT);
    }
    if (p.count_passes && tid == 0 && local_passes) atomicAdd(&p.queue[1], local_passes);
    if (timing) {
#pragma unroll
        for (int i = 0; i < CT_N; ++i) atomicAdd(&p.phase_cycles[i], ph[i]);
    }
#undef PHASE_END
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
    // keys + vals + seg_lo + seg_pre(+64) + seg_v1 + seg_hi + hist + wsum + sh
    return (size_t)T * 8 + (size_t)NT * 4 + (size_t)(NT + 64) * 4 + (size_t)NT * 4 + (size_t)NT * 4 + 256 * 4 + 64 * 4 + 16 * 4;
}

int make_config(const sp_knn_args *a, int n_cus, Config *c) {
    int NT = a->threads_per_wg ? a->threads_per_wg : 512;
    if (NT != 256 && NT != 512 && NT != 1024) return fail(SP_EINVAL, "threads_per_wg must be 256, 512 or 1024 (got %d)", NT);
    int T = a->table_slots ? a->table_slots : 16384;
    if (T < 1024 || (T & (T - 1))) return fail(SP_EINVAL, "table_slots must be a power of two >= 1024 (got %d)", T);
    int logT = 0;
    while ((1 << logT) < T) ++logT;
    const int load = a->load_pct > 0 ? std::min(a->load_pct, 90) : 50;

    const long long need_cap = (long long)a->k + U_SLACK;
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
    kp.phase_cycles = timed ? (unsigned long long *)(ws + 64) : nullptr;   // inside the zeroed queue block
    kp.dbg = (int)a->reserved[0];

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
        unsigned char qb[WS_QUEUE_BYTES];
        HIP_TRY(hipMemcpy(qb, ws, sizeof(qb), hipMemcpyDeviceToHost));
        a->passes_total = (int32_t)((unsigned *)qb)[1];
        const unsigned long long *phc = (const unsigned long long *)(qb + 64);
        for (int i = 0; i < PH_N && i < 6; ++i) a->phase_cycles[i] = (int64_t)phc[i];
        for (int i = 0; i < 4; ++i) a->reserved[i] = (int64_t)phc[PH_N + i];   // debug counters: probe rounds, iterations, overflows, sweeps
        a->num_wgs_used = c.num_wgs;
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
