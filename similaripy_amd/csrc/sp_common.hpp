// sp_common.hpp — device-side vocabulary shared by the kernels of libsimilaripy_hip.so (gfx950 only).
// Epilogue, candidate filter, top-k buffer primitives, selections.  Reference: similaripy/cython_code/s_plus.h.
#pragma once
#ifndef SP_ABLATION
#define SP_ABLATION 0
#endif
#ifndef SP_TRIPTIMERS      // profiling build: wave 0 of every workgroup times its waits for the sweeps' data (phase slots 8 and 11)
#define SP_TRIPTIMERS 0
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/sp_knn.h"

typedef unsigned long long u64;

namespace {


constexpr int EMPTY = -1;                        // key of a free slot (column ids are >= 0)
constexpr u64 EMPTY64 = 0xFFFFFFFF00000000ull;   // free accumulator slot: key EMPTY, partial sum +0.0f
constexpr int MAX_PROBE = 128;   // probe budget of one element before a hashed window is declared overflowed
// m2 elements per lane and trip (two trips are in flight); 1024-thread workgroups have half the VGPR budget
#ifndef ACC_UNROLL
#define ACC_UNROLL (NT >= 1024 ? 2 : NT >= 768 ? 4 : 8)
#endif
constexpr int U_SLACK = 1024;    // candidate-buffer entries beyond k (room between two selections)
// table slots per thread per drain iteration (their Y gathers fly together); 1024-thread workgroups have
// half the VGPR budget (128), where 8 would spill
#ifndef DRAIN_UNROLL
#define DRAIN_UNROLL (NT >= 1024 ? 2 : NT >= 768 ? 4 : 8)
#endif

// scalar slots in LDS
enum { SH_CNT = 0, SH_OVF, SH_SEL, SH_NEED, SH_EQ, SH_CNT2, SH_RETRY, SH_QA, SH_QB, SH_PCTR, SH_MCTR, SH_NITEMS, SH_STOP, SH_WSUM, SH_NHI = SH_WSUM + 16, SH_BINCNT, SH_LIST, SH_N };   // SH_CNT2/SEL/NEED/EQ belong to the selections

// phases timed by lane 0 of every workgroup when KParams::phase_cycles != NULL, then event counters
enum { PH_SETUP = 0, PH_SEGMENTS, PH_ACCUM, PH_DRAIN, PH_SELECT, PH_OUTPUT, PH_SWEEP1, PH_SWEEP2, PH_CSDRAIN,
       CT_ROWS_SPARSE, CT_ROWS_FALLBACK, CT_PASSES, PH_N };

// ---- the bounded variant's per-call facts (written on the device by sp_bnd_range_kernel, read by the row kernel) ----
// W[c] = rho_tv*Ytv[c] + rho_cos*Ycos[c] + rho_dep*Ydep[c] over the live column terms: with per-row multipliers r_j (r_tv = l1*t2,
// r_cos = l2*Xcos[t], r_dep = l3*Xdep[t]) and lam = min_j r_j / rho_j the column part of the denominator obeys
//     sum_j r_j*Y_j[c]  >=  lam*W[c] + sum_j (r_j - lam*rho_j)*ymin_j,
// and W[c] travels with the column id: code(c) = (bits(W[c]) >> 19) - 1 — exponent and four mantissa bits, one quantum down — sits in
// bits 20..31 of every m2 index (n_cols <= 2^20; 21 / 22 id bits and a shorter code up to 2^22 columns, round 6), decoded by ONE shift:  as_float(id >> 1)  <= W[c]  (the id's own bits land below the
// code and add less than the quantum the code was lowered by; the sign bit comes out 0).  The bound is 6 % below W on average, 12.5 % at
// worst.  (Round 5 first carried an adaptive 12-bit code — up to ten mantissa bits, decoded with a shift and an add of per-call
// constants: two more scalar registers and one more instruction per product in the sweep cost more than the tighter bound returned.)
constexpr int BND_ID_BITS = 20;          // ... at least: a call with more output columns takes 21 or 22 (round 6) and gives the code's last mantissa bits up
constexpr int BND_ID_BITS_MAX = 22;      //     — the decode is the same shift: code(c) = (bits(W[c]) >> (id_bits - 1)) - 1 sits at bit id_bits, as_float(id >> 1) <= W[c]
__host__ __device__ constexpr int bnd_id_bits(long long n_cols) { return n_cols <= (1LL << 20) ? 20 : n_cols <= (1LL << 21) ? 21 : 22; }
struct BndInfo {
    int state;             // 1 = usable; anything else: the call runs on the general variant (MODE 0)
    float rho_tv, rho_cos, rho_dep;      // 0 = the term is not part of W (not live, or no usable reference)
    float ymin_tv, ymin_cos, ymin_dep;   // minima over the valid columns (0 where not live)
};
__host__ __device__ inline constexpr size_t bnd_info_bytes() { return sizeof(BndInfo); }

// the column's combined term (one expression, shared by the range and the pack pass; no contraction: both must agree to the bit)
__device__ __forceinline__ float bnd_w(float rtv, float ytv, float rcos, float ycos, float rdep, float ydep) {
    return __fadd_rn(__fadd_rn(__fmul_rn(rtv, ytv), __fmul_rn(rcos, ycos)), __fmul_rn(rdep, ydep));
}
// decode: a lower bound of W[c] from a packed id
__device__ __forceinline__ float bnd_decode(unsigned packed) { return __uint_as_float(packed >> 1); }
// (code = (bits(W) >> (id_bits - 1)) - 1: with 20-bit ids exponent and four mantissa bits, in [1, 4078])
// can a candidate with raw dot x on packed column c still matter?  (nKw = -Kw, see the kernel's set_bnd_cut)
__device__ __forceinline__ bool bnd_alive(unsigned packed, float x, float nKw, float Q) {
    return !(__builtin_fmaf(bnd_decode(packed), nKw, x) <= Q);
}

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
    int T;                 // accumulator slots (power of two); the table region is T*8 bytes
    int logT;
    int cap;               // candidate buffer capacity (> k) of the generic kernel
    int cap_s;             // ... of the sparse kernel
    u64 *gU, *gU_g;        // candidate buffers in global memory (only when they do not fit LDS): sparse / generic kernel
    unsigned int *queue;   // [0] / [1] = next queue position of the sparse / generic kernel (dynamic scheduling)
    unsigned int *qcount;  // [0] / [1] = rows in the sparse / generic queue (the sparse kernel appends its give-ups to [1])
    unsigned int *qcount_g; // rows in the generic queue: where a sparse-row kernel (workgroup- or wave-per-row: each has its own queue, [0] above) appends its give-ups
    const int4 *desc;      // sparse queue: two int4 per row {slot, m1 row, m1 start, m1 length}, {MACs (saturated), Xtv[row], Xcos[row], Xdep[row]}
    int4 *desc_g;          // generic queue, same records
    unsigned m2_bytes;     // nnz(m2) * 4: extent of the m2 index / value buffers (buffer-load range check)
    int nb_log2;           // log2 of the sparse path's column bitmap size in bits (<= log2(T*64))
    int hash_fill;         // slots' worth of MACs one hash window may receive (= T * load_pct / 100)
    int static_sched;
    const float4 *Ypack;   // optional: {Ytv, Ycos, Ydep, 0} per column when two or more column terms are in use (one gather instead of several)
    const float *ymin;     // [3] minima of Ytv / Ycos / Ydep over all columns (valid iff bound_ok)
    int bound_ok;          // weights/shrinks are all >= 0: the epilogue upper bound is sound
    const int *neg_flag;   // Bayesian shrink / Tversky with t1+t2 < 1 only: *neg_flag != 0 iff m1 or m2 holds a negative value (see RowCtx::set_cut)
    int sparse_path;       // 1 = rows with few expected collisions take the bitmap path
    int fold;              // 1 = the single active column term (Ycos or Ydep) is already divided into m2_data: treat it as 1
    // generic kernel, heavy rows cut into pieces — ranges of fine column windows (each piece is a queue entry of its own, any
    // workgroup takes it; sp_merge_pieces_kernel selects the row's top-k from the pieces' results): a queue entry whose slot
    // field is negative is piece number -1 - slot
    const int2 *piece_info;    // [piece] {the row's output slot, first fine window | one past the last << 16}
    int *part_cols;            // [piece * k]
    float *part_vals;          // [piece * k]
    int *part_counts;          // [piece]
    const int *splits;     // generic kernel, optional: [n_splits][splits_rows] (boundary-major) position of the first entry of m2 row u with column >= (j+1)*split_w
    int splits_rows;       //   = n_rows_m2: a heavy row's entries are ascending m2 rows, so one boundary's positions for them lie next to each other
    const int *splits_state;   //   [0] != 0 once the table exists in this workspace: a call whose generic queue holds a handful of rows skips the pass
                               //   over m2 (sp_m2_splits_kernel) and those rows find their window slices by lower_bound, as s_plus.h:385-394 does
    int n_splits;          //   (the boundaries of the fine windows, found once per call instead of once per use)
    int split_w;           //   fine window width: 2T / f, f in {1, 2, 4}; standard dense windows start at multiples of 2T
    // work items of the sparse kernel's rows, cut once per call by sp_row_items_kernel (optional): record 0 of an output slot's
    // block is the header {items, 0, 0, 0} (0 = the row is set up in the kernel), records 1.. are the items of the row
    const int4 *items_g;       // [items_rows][items_stride]
    int items_stride;          // records per output slot: 64, 128 or ITEMS_STRIDE (chosen per call from the average row, sp_knn.hip)
    int items_rows;            // output slots below this have a block
    // bounded variant of the sparse kernel (MODE 2, sp_sparse_kernel.hpp): the m2 column ids with a 12-bit code of the column's
    // combined term W[c] in bits 20..31 (per-call pass), the same packed id per column, and the call's BndInfo
    const unsigned *m2_packed;     // [nnz(m2)]
    const unsigned *colpack;       // [n_cols]
    unsigned bnd_id_mask;          // (1 << id bits of this call) - 1: the column of a packed id
    const struct BndInfo *bnd;     // device memory (workspace header)
    unsigned long long *phase_cycles;  // optional [PH_N]
    int dbg;               // ablation bits for profiling only (results are WRONG when non-zero; 8 / 16: sweep 1 / sweep 2 of the sparse kernel
                           // load but do not process — compiled in only with -DSP_ABLATION=1: the test costs the sweeps 1 %):
                           // 1 = generic accumulate: no LDS inserts, 4 = no Y gathers
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
    float a1, l1, l2, l3, t1, t2, stab, bayes, threshold;
    float xtv, xcos, xdep;  // row terms
    bool any;
    // upper bound without column terms: den >= bA + bB*xy for every column (valid iff bound)
    bool bound;
    bool cut_ok;   // the raw-dot cutoff may be used (false: negative data under a Bayesian shrink or a Tversky term with t1+t2 < 1, see RowCtx::set_cut)
    float bA, bB;

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

    // A value the similarity of a candidate with raw dot xy cannot exceed whatever its column is
    // (+inf when nothing can be said).  Uses only row terms and the per-launch minima of the column
    // terms, so candidates can be discarded before any gather.
    __device__ __forceinline__ float upper(float xy) const {
        if (!any) return xy;                                   // raw dot: exact
        if (!bound) return __builtin_inff();
        const float den = bA + bB * xy;                        // <= true denominator
        if (!(den > 0.f)) return __builtin_inff();
        const float num = (a1 != 1.f) ? powf(xy, a1) : xy;
        if (!(num >= 0.f)) {
            // negative numerator over a positive denominator: the value is negative (NaN stays NaN and is
            // dropped by the threshold test later); only prunable when no Bayesian factor can flip the sign
            return (bayes == 0.f && threshold >= 0.f && num < 0.f) ? -__builtin_inff() : __builtin_inff();
        }
        float v = __fdividef(num, den) * 1.00002f + 1e-30f;    // slack for the few roundings that differ
        return v;                                              // Bayesian factor num/(num+bayes) is <= 1
    }
};

// Keep exactly the k largest of U[0..n) (n > k), in place.  MSD radix-select on the 32-bit key in
// the high half of each entry.  Must be entered by the whole workgroup right after a barrier.
// Returns the key of the k-th largest entry (the new running threshold), or -1 if n <= k (nothing done).
template <int NT>
__device__ long long compact_topk(u64 *U, int *hist, int *sh, int k) {
    const int tid = threadIdx.x;
    const int n = sh[SH_CNT];
    __syncthreads();     // nobody may append (and change SH_CNT) before everyone has read n
    if (n <= k) return -1;  // uniform

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
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (keep) U[wbase + __popcll(m & ((1ull << lane) - 1ull))] = it;
        }
    }
    __syncthreads();
    if (tid == 0) sh[SH_CNT] = sh[SH_CNT2];
    __syncthreads();
    return (long long)prefix;
}

// Top `32 - shift` bits of a multiplicative (Fibonacci) hash of a column id.  A 24-bit multiply would be full
// rate on CDNA but aliases 4-5x more often on uniformly random columns (simulated), which the bitmap path
// pays for directly; one quarter-rate v_mul_lo_u32 per hash is the better trade.
__device__ __forceinline__ unsigned hash_bits(int c, unsigned k, int shift) {
    return ((unsigned)c * k) >> shift;
}

// Row-constant state needed to judge candidates.
struct RowCtx {
    Epi epi;
    int row;               // absolute m1 row id (selector rows are indexed by it, s_plus.h:165-169)
    int f0, f1, g0, g1;    // selector row ranges
    bool have_thr;
    unsigned thr_key;
    float xy_cut;          // a candidate whose raw dot is <= xy_cut cannot enter the top-k (see set_cut)

    // Invert the gather-free upper bound once per (row, running k-th value): the per-product test in the
    // streaming loops becomes ONE float compare.  Conservative: -inf whenever the inversion is not obviously
    // sound, in which case everything stays live and is judged exactly later.
    __device__ __forceinline__ void set_cut(float threshold) {
        const float ninf = -__builtin_inff();
        // the value a candidate must beat: > running k-th value (strict) and >= threshold
        const float below_thr = __uint_as_float(funkey_inv_below(threshold));
        float t = below_thr;
        if (have_thr) t = fmaxf(t, funkey(thr_key));
        xy_cut = ninf;
        if (!epi.any) { xy_cut = t; return; }                        // value == raw dot, exact
        // "raw dot <= cutoff => out" needs the value to grow with the raw dot below the cutoff too.  With NEGATIVE raw
        // dots it does not in two cases: the Bayesian factor xy/(xy+b) turns xy in (-b, 0) into a positive value that
        // grows without bound as xy -> -b; and a Tversky denominator bA + bB*xy with bB > 0 changes sign at
        // xy = -bA/bB, beyond which negative over negative is large and positive.  Both only when a product can be
        // negative at all (flag set per call).
        if (!epi.bound || !epi.cut_ok || epi.a1 != 1.f || !(t >= 0.f) || !(epi.bA > 0.f)) return;
        // ub(xy) = s*xy / (bA + bB*xy) > t   <=>   xy * (s - t*bB) > t*bA      (denominator > 0 region, s = 1.00002)
        const float s = 1.00002f;
        const float d = s - t * epi.bB;
        if (!(d > 0.f)) return;
        xy_cut = (t * epi.bA) / d * 0.99998f;                        // shave: roundings of this formula itself
    }
    // largest float strictly below x, as the order-preserving key mapped back (helper for set_cut)
    static __device__ __forceinline__ unsigned funkey_inv_below(float x) {
        if (!(x == x)) return __float_as_uint(-__builtin_inff());
        unsigned k = fkey(x);
        k = (k == 0u) ? 0u : k - 1u;
        unsigned b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
        if (b == 0x80000000u) b = 0x80000001u;   // -0.0 compares equal to +0.0: step on to the next float below
        return b;
    }
};

// Can a candidate with raw dot xy still enter the top-k?  Gather-free: row terms + per-launch column minima.
__device__ __forceinline__ bool candidate_live(const KParams &p, const RowCtx &rc, float xy) {
    const float ub = rc.epi.upper(xy);
    // NaN bounds compare false on `<` and therefore stay live (the exact path drops them)
    const bool dead = (ub < p.threshold) || (rc.have_thr && fkey(ub) <= rc.thr_key && !(ub != ub));
    return !dead;
}

// Append the items flagged in `mask` (bit j = this lane's item j) to an LDS/global list: lane counts ->
// wave scan -> ONE atomic on the list counter per wave.  store(j, pos) writes item j at list position pos;
// items that do not fit raise *full_flag.  Must be called from wave-uniform control flow.
template <int N, typename Store>
__device__ __forceinline__ void wave_push(unsigned mask, int *counter, int capacity, int *full_flag, Store &&store) {
    // per-item ballots give every lane its rank without any cross-lane data movement (s_bcnt1 / v_mbcnt),
    // the reservation is one returning atomic by lane 0, broadcast with v_readfirstlane
    const int lane = threadIdx.x & 63;
    u64 m[N];
    int off[N];
    int tot = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        m[j] = __ballot((mask >> j) & 1u);
        off[j] = tot;
        tot += __popcll(m[j]);
    }
    if (tot == 0) return;  // wave-uniform
    int wbase = 0;
    if (lane == 0) wbase = atomicAdd(counter, tot);
    wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (mask & (1u << j)) {
            const int pos = wbase + off[j] + __popcll(m[j] & ((1ull << lane) - 1ull));
            if (pos < capacity) store(j, pos); else *full_flag = 1;
        }
    }
}

// Judge N candidates (column c[j], raw dot xy[j]; bit j of `occ` = slot j holds one) held per lane and
// append the survivors to the top-k buffer U.  Order of work: gather-free upper bound -> column selectors
// -> batched gathers of the column terms -> epilogue -> threshold / running k-th value -> one aggregated
// reservation per wave.  Must be called from wave-uniform control flow.
// Returns the candidates that are finished (rejected or stored).  A survivor that finds U full is not in
// the returned mask and SH_RETRY is raised: the caller keeps it and re-offers it after a selection.
// `simple` (wave-uniform): the value depends on the column through the raw dot only — no Tversky term, no column selector,
// the column term folded into m2 or absent (cosine, asymmetric cosine, p3alpha, rp3beta, dot product of the plain calls).  Then
// the row's inverted bound xy_cut is the whole pre-test (one compare per slot) and only survivors see the epilogue.
template <int N>
__device__ __forceinline__ unsigned emit_candidates(const KParams &p, const RowCtx &rc, const int (&c)[N], const float (&xy)[N],
                                                    unsigned occ, u64 *U, int *sh, int cap, bool simple = false) {
    if (simple) {
        unsigned live = 0;
#pragma unroll
        for (int j = 0; j < N; ++j)
            if ((occ & (1u << j)) && !(xy[j] <= rc.xy_cut)) live |= 1u << j;      // (a NaN dot stays live; the exact test drops it)
        if (!__ballot(live != 0)) return occ;
        unsigned want = 0;
        unsigned key[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float val = rc.epi(xy[j], 0.f, 1.f, 1.f);
            key[j] = fkey(val);
            if ((live & (1u << j)) && (val >= p.threshold) && (!rc.have_thr || key[j] > rc.thr_key)) want |= 1u << j;
        }
        unsigned stored = 0;
        wave_push<N>(want, &sh[SH_CNT], cap, &sh[SH_RETRY], [&](int j, int pos) {
            U[pos] = ((u64)key[j] << 32) | (u64)(unsigned)c[j];
            stored |= 1u << j;
        });
        return occ & (~want | stored);
    }
    unsigned live = 0;
#pragma unroll
    for (int j = 0; j < N; ++j)
        if ((occ & (1u << j)) && candidate_live(p, rc, xy[j])) live |= 1u << j;
    if (!__ballot(live != 0)) return occ;  // nothing in this wave can survive: no gathers, no epilogue

    if (p.filter_mode == SP_SEL_MATRIX) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if ((live & (1u << j)) && range_has(p.f_indices, rc.f0, rc.f1, c[j])) live &= ~(1u << j);
    }
    if (p.target_mode == SP_SEL_MATRIX) {
#pragma unroll
        for (int j = 0; j < N; ++j)
            if ((live & (1u << j)) && !range_has(p.t_indices, rc.g0, rc.g1, c[j])) live &= ~(1u << j);
    }
    // gather the column terms of all N candidates first (loads in flight together); dead ones read column 0
    float ytv[N], ycos[N], ydep[N];
    int gc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        gc[j] = ((live & (1u << j)) && !(p.dbg & 4)) ? c[j] : 0;
        ytv[j] = 0.f; ycos[j] = 0.f; ydep[j] = 0.f;
    }
    if (p.Ypack) {
        // two or more column terms: one 16-byte gather per candidate
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float4 y = p.Ypack[gc[j]];
            ytv[j] = y.x; ycos[j] = y.y; ydep[j] = y.z;
        }
    } else {
        if (p.l1 != 0.f) {
#pragma unroll
            for (int j = 0; j < N; ++j) ytv[j] = p.Ytv[gc[j]];
        }
        if (p.l2 != 0.f) {
#pragma unroll
            for (int j = 0; j < N; ++j) ycos[j] = p.fold ? 1.f : p.Ycos[gc[j]];
        }
        if (p.l3 != 0.f) {
#pragma unroll
            for (int j = 0; j < N; ++j) ydep[j] = p.fold ? 1.f : p.Ydep[gc[j]];
        }
    }
    unsigned want = 0;
    unsigned key[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float val = (p.dbg & 128) ? xy[j] : rc.epi(xy[j], ytv[j], ycos[j], ydep[j]);      // (ablation: no epilogue)
        key[j] = fkey(val);
        if ((live & (1u << j)) && (val >= p.threshold) && (!rc.have_thr || key[j] > rc.thr_key)) want |= 1u << j;
    }
    // one aggregated reservation per wave
    unsigned stored = 0;
    if (p.dbg & 64) return occ;                                                                    // (ablation: survivors are dropped, no reservation)
    wave_push<N>(want, &sh[SH_CNT], cap, &sh[SH_RETRY], [&](int j, int pos) {
        U[pos] = ((u64)key[j] << 32) | (u64)(unsigned)c[j];
        stored |= 1u << j;
    });
    return occ & (~want | stored);
}

// ---------------------------------------------------------------------------------------------
// helpers of the sparse path
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x32 __attribute__((ext_vector_type(32)));

constexpr int ITEM = 256;         // m2 elements per work item: one 16-byte load per lane
constexpr int ITEM_CAP = 1008;    // work items per row (LDS: 16 B each) of a 1024-thread workgroup
// ... and of a smaller one: a wave keeps at most 63 item descriptors, so NW*64 entries are all a row can use (the area
// doubles as 4 KiB of scratch for the segment order: never below 256 entries)
__host__ __device__ constexpr int item_cap(int NT) { return NT >= 1024 ? ITEM_CAP : (NT / 64) * 64; }
constexpr int CBM_BYTES = 8192;   // collision bitmap of the sparse kernel (64k bits = 2048 words)
constexpr int PRE_BYTES = 4096;   // per-word exclusive popcount prefix of the collision bitmap (u16 each)
// the two-per-CU shape of the sparse kernel (sp_sparse_kernel.hpp, DUO): 512 threads, 80 KB of LDS —
//   cbm 8 KB | [rank prefix 4 KB | collision set 20 KB | member pool 24 KB | U 16 KB] = the 64 KB (2^19-bit) sweep-1 bitmap | items 3 840 B | histograms 4 KB | scalars 256 B
constexpr int DUO_NT = 512;
constexpr int DUO_CS_DIRECT = 2048;                         // collision-set slots addressed by the rank of a column's mark (a C2 row: ~1 600 marks)
constexpr int DUO_CS_DIRECT_L = 3072, DUO_CS_OVER_L = 1024, DUO_U_ENTRIES_L = 1536;      // ... for calls whose average row expects more marks than that: 32 KB of
                                                            // collision set (the overflow area doubles with the marks: at 82 % full it probes forever), 12 KB of U, a member pool of 2048 entries
constexpr int DUO_CS_OVER = 512;                            // ... and for columns that share a mark or a first-plane bit (~150 per C2 row)
constexpr int DUO_CS_BYTES = (DUO_CS_DIRECT + DUO_CS_OVER) * 8;     // 20 KB
constexpr int DUO_U_BYTES = 16384;                          // candidate buffer: 2048 entries = SEL_E * DUO_NT, what the register-resident selection handles
constexpr int DUO_A_BYTES = 65536 - PRE_BYTES;              // what lies behind the rank prefix
constexpr int DUO_MP_BYTES = DUO_A_BYTES - DUO_CS_BYTES - DUO_U_BYTES;      // member pool: 3072 entries
constexpr int DUO_ICAP = 240;                               // item records (a C2 row has 193)
constexpr int DUO_NB_LOG2 = 19;
__host__ __device__ constexpr size_t sp_duo_lds_bytes() { return (size_t)CBM_BYTES + PRE_BYTES + DUO_A_BYTES + (size_t)DUO_ICAP * 16 + 4096 + 32 * 4 + 16 * 8; }
static_assert(sp_duo_lds_bytes() * 2 <= 160 * 1024, "two workgroups of the DUO shape share a CU's 160 KB");
static_assert(DUO_MP_BYTES >= 8192 && DUO_MP_BYTES % 512 == 0 && (DUO_CS_OVER & (DUO_CS_OVER - 1)) == 0, "DUO layout");
constexpr int POOL_BLK = 64;      // pool entries a wave reserves at a time (>= 64: one trip always fits a fresh block)
constexpr int CS_MAXPROBE = 64;   // linear-probe budget in the collision set
constexpr unsigned OOB_SOFFSET = 0xFFFFF000u;   // buffer-load scalar offset beyond any m2 extent: every lane out of range
constexpr int SEL_E = 4;          // candidate-buffer entries per thread the register-resident selection handles
constexpr int ITEMS_PRE = 255;    // item records per row that the per-call prepass cuts (rows with more, or with more than 64 m1 entries: in the kernel).
                                  // 256 records x 16 B = 4 KB of workspace per output slot (8 KB before; a C2 row has 193 records, a
                                  // user-scoring row 65; rows beyond 255 trips — 65 k products — are set up in the kernel as they were before the prepass)
constexpr int ITEMS_STRIDE = ITEMS_PRE + 1;
// descriptor word .w of a sparse-queue row: m1 entries | trips cut by the prepass << 9 | item records << 19 (0 / 0: cut in the kernel)
__host__ __device__ constexpr int desc_n1(int w) { return w & 0x1FF; }
__host__ __device__ constexpr int desc_n_trips(int w) { return (w >> 9) & 0x3FF; }
__host__ __device__ constexpr int desc_n_rec(int w) { return (int)((unsigned)w >> 19); }
constexpr int ITEM_W_BITS = 20;   // item record .w: products before the trip in the low bits, index of the trip's B record above
constexpr int SORT_MAX = 256;     // m1 rows up to this many entries are visited in descending |value| order

__device__ __forceinline__ int mbcnt64(u64 m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Wave-private window [pos, end) into a shared LDS pool: entries are appended with no atomic at all until
// the window is used up, then ONE returning atomic reserves the next POOL_BLK entries.  Abandoned tails stay
// zero ("hole"); consumers skip zeros.  pos/end are wave-uniform (scalar registers).
struct WavePool { int pos, end; };

// Make room for `tot` (<= 256) more entries in the wave's window: nothing to do while the current block lasts, else
// ONE returning atomic reserves a fresh block (a multiple of POOL_BLK entries).  Returns false when the pool is
// exhausted (flag raised: the row is redone on the generic path).  Wave-uniform.
template <int BLK = POOL_BLK>
__device__ __forceinline__ bool pool_reserve(WavePool &wp, int tot, int *ctr, int cap, int *ovf) {
    // (every read of the window goes through v_readfirstlane: the compiler then keeps pos / end in scalar registers and the
    // test below is a scalar compare and branch — left to itself it held them in vector registers and wrapped the whole push
    // sequence of a sweep body, which runs for nearly every body, in exec-mask saves and restores)
    const int pos = __builtin_amdgcn_readfirstlane(wp.pos), end = __builtin_amdgcn_readfirstlane(wp.end);
    if (pos + tot <= end) return true;
    const int blk = (tot + BLK - 1) & ~(BLK - 1);
    int base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(ctr, blk);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + blk > cap) {
        if ((threadIdx.x & 63) == 0) *ovf = 1;
        wp.pos = 0; wp.end = -1;
        return false;
    }
    wp.pos = base; wp.end = base + blk;
    return true;
}

// ---- hand-scheduled cores of the two sweeps (inline asm: the compiler's own code for these few lines carries two to
// three times the instructions — address re-materialisation, bool <-> mask conversions, redundant masking) ----

// Sweep 1, four columns per lane: OR each column's bit into the bitmap (LDS byte offset BM_OFF, `amask` = byte mask of
// its words) and report whether it was there already.  `one[j]` = 1 for a real element, 0 for padding (ORs nothing).
template <int BM_OFF, bool MASKED>
__device__ __forceinline__ void s1_core(const unsigned (&c)[4], const unsigned (&one)[4], unsigned amask, unsigned (&seen)[4]) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3;
    if (!MASKED) {
        asm volatile(
            "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
            "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
            "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
            "v_lshlrev_b32_e64 %[b0], %[c0], 1\n\t"
            "v_lshlrev_b32_e64 %[b1], %[c1], 1\n\t"
            "v_lshlrev_b32_e64 %[b2], %[c2], 1\n\t"
            "v_lshlrev_b32_e64 %[b3], %[c3], 1\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "v_and_b32 %[a2], %[am], %[a2]\n\t"
            "v_and_b32 %[a3], %[am], %[a3]\n\t"
            "ds_or_rtn_b32 %[b0], %[a0], %[b0] offset:%[off]\n\t"
            "ds_or_rtn_b32 %[b1], %[a1], %[b1] offset:%[off]\n\t"
            "ds_or_rtn_b32 %[b2], %[a2], %[b2] offset:%[off]\n\t"
            "ds_or_rtn_b32 %[b3], %[a3], %[b3] offset:%[off]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_bfe_u32 %[b0], %[b0], %[c0], 1\n\t"
            "v_bfe_u32 %[b1], %[b1], %[c1], 1\n\t"
            "v_bfe_u32 %[b2], %[b2], %[c2], 1\n\t"
            "v_bfe_u32 %[b3], %[b3], %[c3], 1\n\t"
            : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
            : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [am] "s"(amask), [off] "i"(BM_OFF)
            : "memory");
    } else {
        asm volatile(
            "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
            "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
            "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
            "v_lshlrev_b32 %[b0], %[c0], %[o0]\n\t"
            "v_lshlrev_b32 %[b1], %[c1], %[o1]\n\t"
            "v_lshlrev_b32 %[b2], %[c2], %[o2]\n\t"
            "v_lshlrev_b32 %[b3], %[c3], %[o3]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "v_and_b32 %[a2], %[am], %[a2]\n\t"
            "v_and_b32 %[a3], %[am], %[a3]\n\t"
            "ds_or_rtn_b32 %[b0], %[a0], %[b0] offset:%[off]\n\t"
            "ds_or_rtn_b32 %[b1], %[a1], %[b1] offset:%[off]\n\t"
            "ds_or_rtn_b32 %[b2], %[a2], %[b2] offset:%[off]\n\t"
            "ds_or_rtn_b32 %[b3], %[a3], %[b3] offset:%[off]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_bfe_u32 %[b0], %[b0], %[c0], %[o0]\n\t"      // (the field is `one[j]` bits wide: nothing for padding)
            "v_bfe_u32 %[b1], %[b1], %[c1], %[o1]\n\t"
            "v_bfe_u32 %[b2], %[b2], %[c2], %[o2]\n\t"
            "v_bfe_u32 %[b3], %[b3], %[c3], %[o3]\n\t"
            : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3)
            : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [o0] "v"(one[0]), [o1] "v"(one[1]), [o2] "v"(one[2]), [o3] "v"(one[3]),
              [am] "s"(amask), [off] "i"(BM_OFF)
            : "memory");
    }
    seen[0] = b0; seen[1] = b1; seen[2] = b2; seen[3] = b3;
}

// Eight columns (two items) per lane, padding at QUAD granularity: a lane whose quad of item i holds no real element
// (d_i <= 0) ORs nothing and reports nothing — its operand is 0 << c and the width of its v_bfe is 0; ONE compare per item
// instead of one per element.  The lane that holds a partial item's LAST elements treats its whole quad as real: the up to
// three ids behind the item's end (the next m2 row's first ids, or 0 behind the array's end) set bits nobody asked for.
// That is safe — a column marked without a second product only takes the collision-set route, where the sum of its one
// product is exact — and rare (<= 3 of a 256-element item).
// VSH (the DUO shape's aliasing bitmap): the word address is (c >> shift) with shift = 3 + a, i.e. the bitmap index is the column WITHOUT
// its bits 5 .. 4 + a — columns that share a bit differ in one of those bits, hence in the low 16 bits that pick their bit of the
// collision bitmap: the two columns of an aliased pair keep separate marks (modulo aliasing gives both the same mark and the same
// rank slot; measured on C2 rows: the collision set's overflow half 88 % full, 100 k cycles of probing per row).
template <int BM_OFF, bool VSH = false>
__device__ __forceinline__ void s1_core8q(const unsigned (&c)[8], int d0, int d1, unsigned amask, unsigned (&seen)[8], unsigned shift = 3u) {
    unsigned a0, a1, o0, o1;
    if constexpr (VSH) {
    asm volatile(
        "v_cmp_lt_i32 vcc, 0, %[d0]\n\t"
        "v_cndmask_b32 %[o0], 0, 1, vcc\n\t"
        "v_cmp_lt_i32 vcc, 0, %[d1]\n\t"
        "v_cndmask_b32 %[o1], 0, 1, vcc\n\t"
        "v_lshlrev_b32 %[b0], %[c0], %[o0]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c0]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b0], %[a0], %[b0] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b1], %[c1], %[o0]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c1]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b1], %[a1], %[b1] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b2], %[c2], %[o0]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c2]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b2], %[a0], %[b2] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b3], %[c3], %[o0]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c3]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b3], %[a1], %[b3] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b4], %[c4], %[o1]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c4]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b4], %[a0], %[b4] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b5], %[c5], %[o1]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c5]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b5], %[a1], %[b5] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b6], %[c6], %[o1]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c6]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b6], %[a0], %[b6] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b7], %[c7], %[o1]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c7]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b7], %[a1], %[b7] offset:%[off]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[b0], %[b0], %[c0], %[o0]\n\t"
        "v_bfe_u32 %[b1], %[b1], %[c1], %[o0]\n\t"
        "v_bfe_u32 %[b2], %[b2], %[c2], %[o0]\n\t"
        "v_bfe_u32 %[b3], %[b3], %[c3], %[o0]\n\t"
        "v_bfe_u32 %[b4], %[b4], %[c4], %[o1]\n\t"
        "v_bfe_u32 %[b5], %[b5], %[c5], %[o1]\n\t"
        "v_bfe_u32 %[b6], %[b6], %[c6], %[o1]\n\t"
        "v_bfe_u32 %[b7], %[b7], %[c7], %[o1]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [o0] "=&v"(o0), [o1] "=&v"(o1), [b0] "=&v"(seen[0]), [b1] "=&v"(seen[1]), [b2] "=&v"(seen[2]), [b3] "=&v"(seen[3]), [b4] "=&v"(seen[4]), [b5] "=&v"(seen[5]), [b6] "=&v"(seen[6]), [b7] "=&v"(seen[7])
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [d0] "v"(d0), [d1] "v"(d1), [am] "s"(amask), [off] "i"(BM_OFF), [sh] "s"(shift)
        : "memory", "vcc");
    } else {
    asm volatile(
        "v_cmp_lt_i32 vcc, 0, %[d0]\n\t"
        "v_cndmask_b32 %[o0], 0, 1, vcc\n\t"
        "v_cmp_lt_i32 vcc, 0, %[d1]\n\t"
        "v_cndmask_b32 %[o1], 0, 1, vcc\n\t"
        "v_lshlrev_b32 %[b0], %[c0], %[o0]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c0]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b0], %[a0], %[b0] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b1], %[c1], %[o0]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c1]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b1], %[a1], %[b1] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b2], %[c2], %[o0]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c2]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b2], %[a0], %[b2] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b3], %[c3], %[o0]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c3]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b3], %[a1], %[b3] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b4], %[c4], %[o1]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c4]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b4], %[a0], %[b4] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b5], %[c5], %[o1]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c5]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b5], %[a1], %[b5] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b6], %[c6], %[o1]\n\t"
        "v_lshrrev_b32 %[a0], %[sh], %[c6]\n\t"
        "v_and_b32 %[a0], %[am], %[a0]\n\t"
        "ds_or_rtn_b32 %[b6], %[a0], %[b6] offset:%[off]\n\t"
        "v_lshlrev_b32 %[b7], %[c7], %[o1]\n\t"
        "v_lshrrev_b32 %[a1], %[sh], %[c7]\n\t"
        "v_and_b32 %[a1], %[am], %[a1]\n\t"
        "ds_or_rtn_b32 %[b7], %[a1], %[b7] offset:%[off]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[b0], %[b0], %[c0], %[o0]\n\t"
        "v_bfe_u32 %[b1], %[b1], %[c1], %[o0]\n\t"
        "v_bfe_u32 %[b2], %[b2], %[c2], %[o0]\n\t"
        "v_bfe_u32 %[b3], %[b3], %[c3], %[o0]\n\t"
        "v_bfe_u32 %[b4], %[b4], %[c4], %[o1]\n\t"
        "v_bfe_u32 %[b5], %[b5], %[c5], %[o1]\n\t"
        "v_bfe_u32 %[b6], %[b6], %[c6], %[o1]\n\t"
        "v_bfe_u32 %[b7], %[b7], %[c7], %[o1]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [o0] "=&v"(o0), [o1] "=&v"(o1), [b0] "=&v"(seen[0]), [b1] "=&v"(seen[1]), [b2] "=&v"(seen[2]), [b3] "=&v"(seen[3]), [b4] "=&v"(seen[4]), [b5] "=&v"(seen[5]), [b6] "=&v"(seen[6]), [b7] "=&v"(seen[7])
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [d0] "v"(d0), [d1] "v"(d1), [am] "s"(amask), [off] "i"(BM_OFF), [sh] "i"(3)
        : "memory", "vcc");
    }
}

// The same for eight columns (two items) per lane: twice the LDS atomics in flight per wait, address registers reused
// as soon as their atomic is issued.  MASKED: element j of item i is real iff j < d_i (per lane; padding ORs nothing).
template <int BM_OFF, bool MASKED>
__device__ __forceinline__ void s1_core8(const unsigned (&c)[8], int d0, int d1, unsigned amask, unsigned (&seen)[8]) {
    unsigned a0, a1;
    if (!MASKED) {
        asm volatile(
            "v_lshlrev_b32_e64 %[b0], %[c0], 1\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b0], %[a0], %[b0] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b1], %[c1], 1\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b1], %[a1], %[b1] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b2], %[c2], 1\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c2]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b2], %[a0], %[b2] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b3], %[c3], 1\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c3]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b3], %[a1], %[b3] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b4], %[c4], 1\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c4]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b4], %[a0], %[b4] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b5], %[c5], 1\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c5]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b5], %[a1], %[b5] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b6], %[c6], 1\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c6]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b6], %[a0], %[b6] offset:%[off]\n\t"
            "v_lshlrev_b32_e64 %[b7], %[c7], 1\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c7]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b7], %[a1], %[b7] offset:%[off]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_bfe_u32 %[b0], %[b0], %[c0], 1\n\t"
            "v_bfe_u32 %[b1], %[b1], %[c1], 1\n\t"
            "v_bfe_u32 %[b2], %[b2], %[c2], 1\n\t"
            "v_bfe_u32 %[b3], %[b3], %[c3], 1\n\t"
            "v_bfe_u32 %[b4], %[b4], %[c4], 1\n\t"
            "v_bfe_u32 %[b5], %[b5], %[c5], 1\n\t"
            "v_bfe_u32 %[b6], %[b6], %[c6], 1\n\t"
            "v_bfe_u32 %[b7], %[b7], %[c7], 1\n\t"
            : [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(seen[0]), [b1] "=&v"(seen[1]), [b2] "=&v"(seen[2]), [b3] "=&v"(seen[3]), [b4] "=&v"(seen[4]), [b5] "=&v"(seen[5]), [b6] "=&v"(seen[6]), [b7] "=&v"(seen[7])
            : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [am] "s"(amask), [off] "i"(BM_OFF)
            : "memory");
    } else {
        asm volatile(
            "v_cmp_lt_i32 vcc, 0, %[d0]\n\t"
            "v_cndmask_b32 %[b0], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b0], %[c0], %[b0]\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b0], %[a0], %[b0] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 1, %[d0]\n\t"
            "v_cndmask_b32 %[b1], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b1], %[c1], %[b1]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b1], %[a1], %[b1] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 2, %[d0]\n\t"
            "v_cndmask_b32 %[b2], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b2], %[c2], %[b2]\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c2]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b2], %[a0], %[b2] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 3, %[d0]\n\t"
            "v_cndmask_b32 %[b3], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b3], %[c3], %[b3]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c3]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b3], %[a1], %[b3] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 0, %[d1]\n\t"
            "v_cndmask_b32 %[b4], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b4], %[c4], %[b4]\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c4]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b4], %[a0], %[b4] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 1, %[d1]\n\t"
            "v_cndmask_b32 %[b5], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b5], %[c5], %[b5]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c5]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b5], %[a1], %[b5] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 2, %[d1]\n\t"
            "v_cndmask_b32 %[b6], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b6], %[c6], %[b6]\n\t"
            "v_lshrrev_b32 %[a0], 3, %[c6]\n\t"
            "v_and_b32 %[a0], %[am], %[a0]\n\t"
            "ds_or_rtn_b32 %[b6], %[a0], %[b6] offset:%[off]\n\t"
            "v_cmp_lt_i32 vcc, 3, %[d1]\n\t"
            "v_cndmask_b32 %[b7], 0, 1, vcc\n\t"
            "v_lshlrev_b32 %[b7], %[c7], %[b7]\n\t"
            "v_lshrrev_b32 %[a1], 3, %[c7]\n\t"
            "v_and_b32 %[a1], %[am], %[a1]\n\t"
            "ds_or_rtn_b32 %[b7], %[a1], %[b7] offset:%[off]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_bfe_u32 %[b0], %[b0], %[c0], 1\n\t"
            "v_cmp_lt_i32 vcc, 0, %[d0]\n\t"
            "v_cndmask_b32 %[b0], 0, %[b0], vcc\n\t"
            "v_bfe_u32 %[b1], %[b1], %[c1], 1\n\t"
            "v_cmp_lt_i32 vcc, 1, %[d0]\n\t"
            "v_cndmask_b32 %[b1], 0, %[b1], vcc\n\t"
            "v_bfe_u32 %[b2], %[b2], %[c2], 1\n\t"
            "v_cmp_lt_i32 vcc, 2, %[d0]\n\t"
            "v_cndmask_b32 %[b2], 0, %[b2], vcc\n\t"
            "v_bfe_u32 %[b3], %[b3], %[c3], 1\n\t"
            "v_cmp_lt_i32 vcc, 3, %[d0]\n\t"
            "v_cndmask_b32 %[b3], 0, %[b3], vcc\n\t"
            "v_bfe_u32 %[b4], %[b4], %[c4], 1\n\t"
            "v_cmp_lt_i32 vcc, 0, %[d1]\n\t"
            "v_cndmask_b32 %[b4], 0, %[b4], vcc\n\t"
            "v_bfe_u32 %[b5], %[b5], %[c5], 1\n\t"
            "v_cmp_lt_i32 vcc, 1, %[d1]\n\t"
            "v_cndmask_b32 %[b5], 0, %[b5], vcc\n\t"
            "v_bfe_u32 %[b6], %[b6], %[c6], 1\n\t"
            "v_cmp_lt_i32 vcc, 2, %[d1]\n\t"
            "v_cndmask_b32 %[b6], 0, %[b6], vcc\n\t"
            "v_bfe_u32 %[b7], %[b7], %[c7], 1\n\t"
            "v_cmp_lt_i32 vcc, 3, %[d1]\n\t"
            "v_cndmask_b32 %[b7], 0, %[b7], vcc\n\t"
            : [a0] "=&v"(a0), [a1] "=&v"(a1), [b0] "=&v"(seen[0]), [b1] "=&v"(seen[1]), [b2] "=&v"(seen[2]), [b3] "=&v"(seen[3]), [b4] "=&v"(seen[4]), [b5] "=&v"(seen[5]), [b6] "=&v"(seen[6]), [b7] "=&v"(seen[7])
            : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]), [d0] "v"(d0), [d1] "v"(d1), [am] "s"(amask), [off] "i"(BM_OFF)
            : "memory", "vcc");
    }
}

// Sweep 2, four products per lane: x = value * segv (segv per lane: the two pieces of a packed trip belong to different m1
// entries), M[j] = lanes whose column is marked in the collision bitmap
// (LDS offset 0, CBM_BYTES long), L[j] = lanes with !(x <= cut)  (NaN counts as above the cutoff).
__device__ __forceinline__ void s2_core(const unsigned (&c)[4], const float (&v)[4], float segv, float cut, float (&x)[4], u64 (&M)[4], u64 (&L)[4]) {
    unsigned a0, a1, a2, a3;
    static_assert(CBM_BYTES == 8192, "the literal below is CBM_BYTES - 4");
    asm volatile(
        "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
        "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
        "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
        "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
        "v_and_b32 %[a0], 0x1ffc, %[a0]\n\t"
        "v_and_b32 %[a1], 0x1ffc, %[a1]\n\t"
        "v_and_b32 %[a2], 0x1ffc, %[a2]\n\t"
        "v_and_b32 %[a3], 0x1ffc, %[a3]\n\t"
        "ds_read_b32 %[a0], %[a0]\n\t"
        "ds_read_b32 %[a1], %[a1]\n\t"
        "ds_read_b32 %[a2], %[a2]\n\t"
        "ds_read_b32 %[a3], %[a3]\n\t"
        "v_mul_f32 %[x0], %[sv], %[v0]\n\t"
        "v_mul_f32 %[x1], %[sv], %[v1]\n\t"
        "v_mul_f32 %[x2], %[sv], %[v2]\n\t"
        "v_mul_f32 %[x3], %[sv], %[v3]\n\t"
        "v_cmp_nle_f32_e64 %[L0], %[x0], %[cut]\n\t"
        "v_cmp_nle_f32_e64 %[L1], %[x1], %[cut]\n\t"
        "v_cmp_nle_f32_e64 %[L2], %[x2], %[cut]\n\t"
        "v_cmp_nle_f32_e64 %[L3], %[x3], %[cut]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[a0], %[a0], %[c0], 1\n\t"
        "v_bfe_u32 %[a1], %[a1], %[c1], 1\n\t"
        "v_bfe_u32 %[a2], %[a2], %[c2], 1\n\t"
        "v_bfe_u32 %[a3], %[a3], %[c3], 1\n\t"
        "v_cmp_ne_u32_e64 %[M0], 0, %[a0]\n\t"
        "v_cmp_ne_u32_e64 %[M1], 0, %[a1]\n\t"
        "v_cmp_ne_u32_e64 %[M2], 0, %[a2]\n\t"
        "v_cmp_ne_u32_e64 %[M3], 0, %[a3]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),
          [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]),
          [M0] "=&s"(M[0]), [M1] "=&s"(M[1]), [M2] "=&s"(M[2]), [M3] "=&s"(M[3]),
          [L0] "=&s"(L[0]), [L1] "=&s"(L[1]), [L2] "=&s"(L[2]), [L3] "=&s"(L[3])
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]),
          [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [sv] "v"(segv), [cut] "s"(cut)
        : "memory");
}

// Sweep 2 of the bounded variant (MODE 2): the same, but the per-product test is  !(x - Kw*W(c) <= Q)  with W decoded from the code in
// the id's upper bits (one shift) — two instructions per product more than the monotone variant's compare.
__device__ __forceinline__ void s2_core_b(const unsigned (&c)[4], const float (&v)[4], float segv, float nKw, float Q,
                                          float (&x)[4], u64 (&M)[4], u64 (&L)[4]) {
    unsigned a0, a1, a2, a3, t0, t1;
    static_assert(CBM_BYTES == 8192, "the literal below is CBM_BYTES - 4");
    asm volatile(
        "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
        "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
        "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
        "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
        "v_and_b32 %[a0], 0x1ffc, %[a0]\n\t"
        "v_and_b32 %[a1], 0x1ffc, %[a1]\n\t"
        "v_and_b32 %[a2], 0x1ffc, %[a2]\n\t"
        "v_and_b32 %[a3], 0x1ffc, %[a3]\n\t"
        "ds_read_b32 %[a0], %[a0]\n\t"
        "ds_read_b32 %[a1], %[a1]\n\t"
        "ds_read_b32 %[a2], %[a2]\n\t"
        "ds_read_b32 %[a3], %[a3]\n\t"
        "v_mul_f32 %[x0], %[sv], %[v0]\n\t"
        "v_mul_f32 %[x1], %[sv], %[v1]\n\t"
        "v_mul_f32 %[x2], %[sv], %[v2]\n\t"
        "v_mul_f32 %[x3], %[sv], %[v3]\n\t"
        "v_lshrrev_b32 %[t0], 1, %[c0]\n\t"
        "v_lshrrev_b32 %[t1], 1, %[c1]\n\t"
        "v_fma_f32 %[t0], %[t0], %[nkw], %[x0]\n\t"
        "v_fma_f32 %[t1], %[t1], %[nkw], %[x1]\n\t"
        "v_cmp_nle_f32_e64 %[L0], %[t0], %[q]\n\t"
        "v_cmp_nle_f32_e64 %[L1], %[t1], %[q]\n\t"
        "v_lshrrev_b32 %[t0], 1, %[c2]\n\t"
        "v_lshrrev_b32 %[t1], 1, %[c3]\n\t"
        "v_fma_f32 %[t0], %[t0], %[nkw], %[x2]\n\t"
        "v_fma_f32 %[t1], %[t1], %[nkw], %[x3]\n\t"
        "v_cmp_nle_f32_e64 %[L2], %[t0], %[q]\n\t"
        "v_cmp_nle_f32_e64 %[L3], %[t1], %[q]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[a0], %[a0], %[c0], 1\n\t"
        "v_bfe_u32 %[a1], %[a1], %[c1], 1\n\t"
        "v_bfe_u32 %[a2], %[a2], %[c2], 1\n\t"
        "v_bfe_u32 %[a3], %[a3], %[c3], 1\n\t"
        "v_cmp_ne_u32_e64 %[M0], 0, %[a0]\n\t"
        "v_cmp_ne_u32_e64 %[M1], 0, %[a1]\n\t"
        "v_cmp_ne_u32_e64 %[M2], 0, %[a2]\n\t"
        "v_cmp_ne_u32_e64 %[M3], 0, %[a3]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]),
          [M0] "=&s"(M[0]), [M1] "=&s"(M[1]), [M2] "=&s"(M[2]), [M3] "=&s"(M[3]),
          [L0] "=&s"(L[0]), [L1] "=&s"(L[1]), [L2] "=&s"(L[2]), [L3] "=&s"(L[3])
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]),
          [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [sv] "v"(segv),
          [nkw] "s"(nKw), [q] "s"(Q)
        : "memory");
}

// ---- DUO: the collision bitmap as TWO planes of 32 k bits (LDS [0, 4096) and [4096, 8192)) ----
// A column c owns bit (c & 31) of word (c >> 5) & 1023 in the first plane and bit ((c >> 15) & 31) of the SAME word index in the second;
// marking sets both, "marked" means both are set.  With one plane of 64 k bits a C2 row in the two-per-CU shape (1.6 k marks: aliased
// pairs mark like repeated columns) has 41 k * 1600 / 65536 = 1 000 products that hit another column's mark; every one of them is carried
// through the member pool into the collision set, where they fill the overflow half (measured: 33 k cycles of accumulate per row, most of
// it probing).  Two planes: (1600 / 32768)^2 = 0.24 % = ~100.  Ranks (the collision set's direct slots) count first-plane bits only.
constexpr int DUO_PLANE_BYTES = CBM_BYTES / 2;
__device__ __forceinline__ void duo_mark(unsigned char *cbm, unsigned c) {
    const unsigned w = (c >> 3) & (unsigned)(DUO_PLANE_BYTES - 4);
    atomicOr((unsigned *)(cbm + w), 1u << (c & 31u));
    atomicOr((unsigned *)(cbm + DUO_PLANE_BYTES + w), 1u << ((c >> 15) & 31u));
}
__device__ __forceinline__ void duo_unmark(unsigned char *cbm, unsigned c) {
    const unsigned w = (c >> 3) & (unsigned)(DUO_PLANE_BYTES - 4);
    atomicAnd((unsigned *)(cbm + w), ~(1u << (c & 31u)));
    atomicAnd((unsigned *)(cbm + DUO_PLANE_BYTES + w), ~(1u << ((c >> 15) & 31u)));
}
// s2_core on the two planes: two reads from one address register, the second bit extracted with the first as its WIDTH (0: not marked)
__device__ __forceinline__ void s2_core_duo(const unsigned (&c)[4], const float (&v)[4], float segv, float cut, float (&x)[4], u64 (&M)[4], u64 (&L)[4]) {
    unsigned a0, a1, a2, a3, h0, h1, h2, h3, u;
    static_assert(DUO_PLANE_BYTES == 4096, "the literals below are DUO_PLANE_BYTES - 4 and DUO_PLANE_BYTES");
    asm volatile(
        "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
        "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
        "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
        "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
        "v_and_b32 %[a0], 0xffc, %[a0]\n\t"
        "v_and_b32 %[a1], 0xffc, %[a1]\n\t"
        "v_and_b32 %[a2], 0xffc, %[a2]\n\t"
        "v_and_b32 %[a3], 0xffc, %[a3]\n\t"
        "ds_read_b32 %[h0], %[a0] offset:4096\n\t"
        "ds_read_b32 %[h1], %[a1] offset:4096\n\t"
        "ds_read_b32 %[h2], %[a2] offset:4096\n\t"
        "ds_read_b32 %[h3], %[a3] offset:4096\n\t"
        "ds_read_b32 %[a0], %[a0]\n\t"
        "ds_read_b32 %[a1], %[a1]\n\t"
        "ds_read_b32 %[a2], %[a2]\n\t"
        "ds_read_b32 %[a3], %[a3]\n\t"
        "v_mul_f32 %[x0], %[sv], %[v0]\n\t"
        "v_mul_f32 %[x1], %[sv], %[v1]\n\t"
        "v_mul_f32 %[x2], %[sv], %[v2]\n\t"
        "v_mul_f32 %[x3], %[sv], %[v3]\n\t"
        "v_cmp_nle_f32_e64 %[L0], %[x0], %[cut]\n\t"
        "v_cmp_nle_f32_e64 %[L1], %[x1], %[cut]\n\t"
        "v_cmp_nle_f32_e64 %[L2], %[x2], %[cut]\n\t"
        "v_cmp_nle_f32_e64 %[L3], %[x3], %[cut]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[a0], %[a0], %[c0], 1\n\t"
        "v_lshrrev_b32 %[u], 15, %[c0]\n\t"
        "v_bfe_u32 %[a1], %[a1], %[c1], 1\n\t"
        "v_bfe_u32 %[a0], %[h0], %[u], %[a0]\n\t"
        "v_lshrrev_b32 %[u], 15, %[c1]\n\t"
        "v_bfe_u32 %[a2], %[a2], %[c2], 1\n\t"
        "v_bfe_u32 %[a1], %[h1], %[u], %[a1]\n\t"
        "v_lshrrev_b32 %[u], 15, %[c2]\n\t"
        "v_bfe_u32 %[a3], %[a3], %[c3], 1\n\t"
        "v_bfe_u32 %[a2], %[h2], %[u], %[a2]\n\t"
        "v_lshrrev_b32 %[u], 15, %[c3]\n\t"
        "v_cmp_ne_u32_e64 %[M0], 0, %[a0]\n\t"
        "v_bfe_u32 %[a3], %[h3], %[u], %[a3]\n\t"
        "v_cmp_ne_u32_e64 %[M1], 0, %[a1]\n\t"
        "v_cmp_ne_u32_e64 %[M2], 0, %[a2]\n\t"
        "v_cmp_ne_u32_e64 %[M3], 0, %[a3]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [u] "=&v"(u),
          [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]),
          [M0] "=&s"(M[0]), [M1] "=&s"(M[1]), [M2] "=&s"(M[2]), [M3] "=&s"(M[3]),
          [L0] "=&s"(L[0]), [L1] "=&s"(L[1]), [L2] "=&s"(L[2]), [L3] "=&s"(L[3])
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]),
          [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [sv] "v"(segv), [cut] "s"(cut)
        : "memory");
}
// ... and the bounded variant's (s2_core_b)
__device__ __forceinline__ void s2_core_b_duo(const unsigned (&c)[4], const float (&v)[4], float segv, float nKw, float Q,
                                              float (&x)[4], u64 (&M)[4], u64 (&L)[4]) {
    unsigned a0, a1, a2, a3, h0, h1, h2, h3, t0, t1;
    asm volatile(
        "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
        "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
        "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
        "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
        "v_and_b32 %[a0], 0xffc, %[a0]\n\t"
        "v_and_b32 %[a1], 0xffc, %[a1]\n\t"
        "v_and_b32 %[a2], 0xffc, %[a2]\n\t"
        "v_and_b32 %[a3], 0xffc, %[a3]\n\t"
        "ds_read_b32 %[h0], %[a0] offset:4096\n\t"
        "ds_read_b32 %[h1], %[a1] offset:4096\n\t"
        "ds_read_b32 %[h2], %[a2] offset:4096\n\t"
        "ds_read_b32 %[h3], %[a3] offset:4096\n\t"
        "ds_read_b32 %[a0], %[a0]\n\t"
        "ds_read_b32 %[a1], %[a1]\n\t"
        "ds_read_b32 %[a2], %[a2]\n\t"
        "ds_read_b32 %[a3], %[a3]\n\t"
        "v_mul_f32 %[x0], %[sv], %[v0]\n\t"
        "v_mul_f32 %[x1], %[sv], %[v1]\n\t"
        "v_mul_f32 %[x2], %[sv], %[v2]\n\t"
        "v_mul_f32 %[x3], %[sv], %[v3]\n\t"
        "v_lshrrev_b32 %[t0], 1, %[c0]\n\t"
        "v_lshrrev_b32 %[t1], 1, %[c1]\n\t"
        "v_fma_f32 %[t0], %[t0], %[nkw], %[x0]\n\t"
        "v_fma_f32 %[t1], %[t1], %[nkw], %[x1]\n\t"
        "v_cmp_nle_f32_e64 %[L0], %[t0], %[q]\n\t"
        "v_cmp_nle_f32_e64 %[L1], %[t1], %[q]\n\t"
        "v_lshrrev_b32 %[t0], 1, %[c2]\n\t"
        "v_lshrrev_b32 %[t1], 1, %[c3]\n\t"
        "v_fma_f32 %[t0], %[t0], %[nkw], %[x2]\n\t"
        "v_fma_f32 %[t1], %[t1], %[nkw], %[x3]\n\t"
        "v_cmp_nle_f32_e64 %[L2], %[t0], %[q]\n\t"
        "v_cmp_nle_f32_e64 %[L3], %[t1], %[q]\n\t"
        "v_lshrrev_b32 %[t0], 15, %[c0]\n\t"
        "v_lshrrev_b32 %[t1], 15, %[c1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[a0], %[a0], %[c0], 1\n\t"
        "v_bfe_u32 %[a1], %[a1], %[c1], 1\n\t"
        "v_bfe_u32 %[a0], %[h0], %[t0], %[a0]\n\t"
        "v_bfe_u32 %[a1], %[h1], %[t1], %[a1]\n\t"
        "v_lshrrev_b32 %[t0], 15, %[c2]\n\t"
        "v_lshrrev_b32 %[t1], 15, %[c3]\n\t"
        "v_bfe_u32 %[a2], %[a2], %[c2], 1\n\t"
        "v_bfe_u32 %[a3], %[a3], %[c3], 1\n\t"
        "v_bfe_u32 %[a2], %[h2], %[t0], %[a2]\n\t"
        "v_bfe_u32 %[a3], %[h3], %[t1], %[a3]\n\t"
        "v_cmp_ne_u32_e64 %[M0], 0, %[a0]\n\t"
        "v_cmp_ne_u32_e64 %[M1], 0, %[a1]\n\t"
        "v_cmp_ne_u32_e64 %[M2], 0, %[a2]\n\t"
        "v_cmp_ne_u32_e64 %[M3], 0, %[a3]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3),
          [t0] "=&v"(t0), [t1] "=&v"(t1),
          [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]),
          [M0] "=&s"(M[0]), [M1] "=&s"(M[1]), [M2] "=&s"(M[2]), [M3] "=&s"(M[3]),
          [L0] "=&s"(L[0]), [L1] "=&s"(L[1]), [L2] "=&s"(L[2]), [L3] "=&s"(L[3])
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]),
          [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [sv] "v"(segv),
          [nkw] "s"(nKw), [q] "s"(Q)
        : "memory");
}

// per lane: `if_set` where the lane's bit of the wave-uniform mask `m` is set, else `if_clear` — ONE v_cndmask on the mask
// in scalar registers (the compiler's form of `(m >> lane) & 1 ? a : b` is a 64-bit shift, an and and a compare in front of it)
__device__ __forceinline__ unsigned mask_select(u64 m_in, unsigned if_set, unsigned if_clear) {
    const u64 m = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m_in >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m_in);
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}

// Append the 8-byte entries {lo, hi} of the lanes in `m` to an LDS list at byte offset `base_off` (wave-uniform),
// starting at entry `pos` (wave-uniform): lane rank by v_mbcnt, exec narrowed to `m` around the two stores.
__device__ __forceinline__ void lds_push64(u64 m_in, unsigned lo, unsigned hi, int pos, unsigned base_off) {
    unsigned t;
    u64 sv;
    // (the mask is wave-uniform, but under register pressure the compiler may keep it in vector registers, which an "s"
    // operand cannot take: readfirstlane pins it to scalar registers and folds away when it is there already)
    const u64 m = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m_in >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m_in);
    // byte address of the list's entry `pos`: scalar arithmetic (one SALU instruction), so that a lane's own address is ONE
    // v_lshl_add on its rank; both words go out in one ds_write2_b32
    const unsigned first = base_off + ((unsigned)__builtin_amdgcn_readfirstlane(pos) << 3);
    asm volatile(
        "v_mbcnt_lo_u32_b32 %[t], %[mlo], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[t], %[mhi], %[t]\n\t"
        "v_lshl_add_u32 %[t], %[t], 3, %[first]\n\t"
        "s_and_saveexec_b64 %[sv], %[m]\n\t"
        "ds_write2_b32 %[t], %[lo], %[hi] offset1:1\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        : [t] "=&v"(t), [sv] "=&s"(sv)
        : [mlo] "s"((unsigned)m), [mhi] "s"((unsigned)(m >> 32)), [m] "s"(m), [first] "s"(first), [lo] "v"(lo), [hi] "v"(hi)
        : "memory", "scc");      // (s_and_saveexec writes SCC: without the clobber the compiler kept a compare's result across a push)
}

// inclusive wave64 scan on the DPP crossbar (row_shr 1/2/4/8, row_bcast 15/31): no LDS round trips
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// wave64 maximum of an unsigned value on the DPP crossbar; the result is wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));    // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));    // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));    // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));    // row_shr:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));    // row_bcast:15
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Workgroup barrier of the row kernels.  LDS_ONLY: waits for this wave's LDS operations only — not for its global loads,
// which the register-resident row kernel keeps in flight across whole phases (__syncthreads() is a workgroup-scope
// fence and would drain vmcnt too); everything the waves of a row exchange goes through LDS.
template <bool LDS_ONLY>
__device__ __forceinline__ void wg_sync() {
    if constexpr (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}

// Selection for candidate buffers of at most E*NT entries: every thread keeps its (<= E) entries in registers,
// one LDS histogram per radix pass (hist4 = 4 x 256 counters, zero on entry and on exit), every wave scans the
// histogram redundantly (no broadcast barrier), 1 barrier per pass.
//   exact:  keeps exactly k entries; returns the key of the k-th largest.
//   !exact: stops after two passes (sign, exponent, 7 mantissa bits) when that already removes most of the
//           surplus: keeps every entry >= the lower edge of the 16-bit bin holding the k-th largest and returns
//           that edge — a valid (conservative) running cutoff, cheaper than the exact one.
// Must be entered by the whole workgroup.  Returns -1 when n <= k (nothing done).
// ZERO_TAIL: the buffer is used with block-wise reservations (holes = zero entries): positions beyond the kept
// entries are zeroed so that a later reservation never exposes stale entries.
// hint: a key every live entry is known to reach (the running cutoff; 0 = none).  An exact selection then starts at the
// second digit: the entries whose first byte exceeds the hint's are only counted, the others of its byte go straight
// into the second histogram — one radix pass (fill, two barriers, walk) less whenever the k-th largest shares the
// hint's first byte, which is the rule (else: the regular four passes).
// CLEAN: the caller's kernel uses SH_CNT2 / SH_EQ / SH_NHI through this function only, which then leaves them at zero on its way
// out instead of resetting them (+ a barrier) on its way in.
template <int NT, bool ZERO_TAIL = false, int E = 4, bool LDSBAR = false, bool CLEAN = false>
__device__ long long select_fast(u64 *U, int *hist4, int *sh, int k, bool exact, unsigned hint = 0u, int tid_in = -1) {
    // (tid_in: a caller that keeps its thread id opaque to stop address computations from being hoisted out of its row loop)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63;
    const int n = min(sh[SH_CNT], E * NT);
    if (!CLEAN) {
        if (tid == 0) { sh[SH_CNT2] = 0; sh[SH_EQ] = 0; sh[SH_NHI] = 0; }   // (the generic path's selection leaves them dirty)
        wg_sync<LDSBAR>();
        if (n <= k) return -1;
    } else if (n <= k) {      // (uniform; rare: callers ask for a selection when there is something to select)
        wg_sync<LDSBAR>();    // nobody may append (and change SH_CNT) before everyone has read n
        return -1;
    }
    u64 e[E];
    unsigned key[E];
    bool has[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int i = tid + j * NT;
        has[j] = i < n;
        e[j] = has[j] ? U[i] : 0ull;
        key[j] = (unsigned)(e[j] >> 32);
    }
    unsigned prefix = 0;
    int need = k;
    int passes = 0;
    int first_ps = 0;
    bool prefilled = false;
    if (exact && hint != 0u) {       // (uniform; intermediate selections measured slower with it: their cutoff is often a byte below)
        const unsigned top = hint >> 24;
        int *h1 = hist4 + 256;
        int mine = 0;                // entries above the hint's first byte | entries of that byte << 16
#pragma unroll
        for (int j = 0; j < E; ++j) {
            if (has[j]) {
                const unsigned kb = key[j] >> 24;
                if (kb == top) { atomicAdd(&h1[(key[j] >> 16) & 255u], 1); mine += 1 << 16; }
                else if (kb > top) mine += 1;
            }
        }
        const int tot = wave_incl_scan_dpp(mine);
        if (lane == 63 && tot) atomicAdd(&sh[SH_NHI], tot);
        wg_sync<LDSBAR>();
        const int both = sh[SH_NHI];
        const int n_hi = both & 0xFFFF, n_top = both >> 16;
        if (n_hi < k && k - n_hi <= n_top) {
            prefix = top << 24;
            need = k - n_hi;
            first_ps = 1;
            prefilled = true;
            passes = 1;
        } else {                     // the k-th largest is not in the hint's byte: regular passes
            for (int i = tid; i < 256; i += NT) h1[i] = 0;
            wg_sync<LDSBAR>();
        }
    }
    for (int ps = first_ps; ps < 4; ++ps) {
        const int shift = 24 - 8 * ps;
        const unsigned hmask = (ps == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
        int *h = hist4 + ps * 256;
        if (!(prefilled && ps == 1)) {
#pragma unroll
            for (int j = 0; j < E; ++j)
                if (has[j] && ((key[j] ^ prefix) & hmask) == 0u) atomicAdd(&h[(key[j] >> shift) & 255u], 1);
            wg_sync<LDSBAR>();
        }
        // ONE wave walks the histogram (the others would only repeat the same instructions) and publishes the digit
        if (tid < 64) {
            // lane L owns bins 255-4L .. 252-4L (lanes ascend as digits descend)
            const int4 c4 = *(const int4 *)&h[252 - 4 * lane];
            const int c0 = c4.w, c1 = c4.z, c2 = c4.y, c3 = c4.x;
            const int s = c0 + c1 + c2 + c3;
            const int incl = wave_incl_scan_dpp(s);
            const int excl = incl - s;
            if (excl < need && need <= incl) {         // exactly one lane
                const int b0 = 255 - 4 * lane;
                int r = need - excl, d, cb;
                if (r <= c0) { d = b0; cb = c0; }
                else if (r <= c0 + c1) { d = b0 - 1; r -= c0; cb = c1; }
                else if (r <= c0 + c1 + c2) { d = b0 - 2; r -= c0 + c1; cb = c2; }
                else { d = b0 - 3; r -= c0 + c1 + c2; cb = c3; }
                // !exact: keeping the whole bin leaves (k - r) + cb entries: good enough after the second pass when that
                // removes at least half of the surplus
                const int kept = (k - r) + cb;
                sh[SH_SEL] = d;
                sh[SH_NEED] = r;
                sh[SH_BINCNT] = cb;
                sh[SH_STOP] = (!exact && ps == 1 && 2 * (kept - k) <= (n - k)) ? 1 : 0;
            }
        }
        wg_sync<LDSBAR>();
        prefix |= (unsigned)sh[SH_SEL] << shift;
        need = sh[SH_NEED];
        passes = ps + 1;
        if (sh[SH_STOP]) break;     // uniform
        if (exact && ps < 3) {
            // The k-th largest is one of the `cb` entries of the bin just chosen.  Once that is at most a wave's worth (after the
            // second byte the rule: a few hundred entries over 65 536 prefixes) the remaining bytes need no histogram passes: the
            // bin's keys are gathered in a list, ONE wave ranks them against each other and publishes the full key of the need-th
            // largest — two barriers instead of two per remaining byte.
            const int cb = sh[SH_BINCNT];
            if (cb <= 64) {     // uniform
                unsigned *lst = (unsigned *)(hist4 + 3 * 256);      // the last pass's histogram: not needed any more (zeroed again below)
                const unsigned bmask = 0xFFFFFFFFu << shift;
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (has[j] && ((key[j] ^ prefix) & bmask) == 0u) lst[atomicAdd(&sh[SH_LIST], 1)] = key[j];
                wg_sync<LDSBAR>();
                if (tid < 64) {
                    const unsigned mine = (lane < cb) ? lst[lane] : 0u;
                    int gt = 0, eq_before = 0;
                    for (int j = 0; j < cb; ++j) {
                        const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)mine, j);
                        gt += (kj > mine) ? 1 : 0;
                        eq_before += (kj == mine && j < lane) ? 1 : 0;
                    }
                    if (lane < cb && gt + eq_before == need - 1) {      // exactly one lane
                        sh[SH_SEL] = (int)mine;
                        sh[SH_NEED] = need - gt;                         // of the entries equal to it, this many are kept
                        sh[SH_LIST] = 0;
                    }
                }
                wg_sync<LDSBAR>();
                prefix = (unsigned)sh[SH_SEL];
                need = sh[SH_NEED];
                passes = 4;
                break;
            }
        }
    }
    const bool all_passes = (passes == 4);
#pragma unroll
    for (int j = 0; j < E; ++j) {
        bool keep = false;
        if (has[j]) {
            if (key[j] > prefix) keep = true;
            else if (key[j] == prefix) keep = all_passes ? (atomicAdd(&sh[SH_EQ], 1) < need) : true;
            else if (!all_passes) keep = (key[j] >= prefix);     // prefix has its low bits clear: the bin's lower edge
        }
        const u64 m = __ballot(keep);
        if (m) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&sh[SH_CNT2], __popcll(m));
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (keep) U[wbase + mbcnt64(m)] = e[j];
        }
    }
    wg_sync<LDSBAR>();
    if (ZERO_TAIL) {
        const int kept = sh[SH_CNT2];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int i = tid + j * NT;
            if (has[j] && i >= kept) U[i] = 0ull;
        }
    }
    for (int i = tid; i < passes * 256; i += NT) hist4[i] = 0;
    if (tid == 0) sh[SH_CNT] = sh[SH_CNT2];
    wg_sync<LDSBAR>();
    // (BEHIND the barrier: every thread reads SH_CNT2 above — zeroed in front of it, a slower wave saw 0 kept entries and wiped its
    // share of the buffer; found by repeating a fuzz seed.  The next reader of the three counters is the next selection, at least one
    // barrier of the caller away)
    if (CLEAN && tid == 0) { sh[SH_CNT2] = 0; sh[SH_EQ] = 0; sh[SH_NHI] = 0; }
    return (long long)prefix;
}


}  // namespace
