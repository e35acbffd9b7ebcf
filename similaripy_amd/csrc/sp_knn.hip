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
//   * persistent workgroups (one per CU) pull target rows from an atomic queue — the analogue of
//     `omp for schedule(dynamic)` (s_plus.h:337);
//   * m2 rows are streamed with lane-contiguous (coalesced) index/value loads: the nnz1(t) row
//     slices are flattened through an LDS prefix array so all 64 lanes stay busy whatever the slice
//     lengths, ACC_UNROLL loads per lane in flight;
//   * the per-thread dense `sums[]` array of the reference (n_cols*4 B, cache-hostile) becomes LDS
//     state, in one of two shapes chosen per row from MACs(t) = sum_u nnz(m2 row u):
//       SPARSE rows (few products share a column — the recommender/KNN shape the headline benchmark
//       has): two sweeps over the row's products.  Sweep 1 sets one hashed bit per column in an LDS
//       bitmap (ds_or_rtn_b32); a product that finds its bit already set enters its column in a small
//       bucketed "collision set".  Sweep 2 looks every product up in that set with ONE ds_read_b128:
//       members accumulate there, every other product is provably the only one of its column and goes
//       straight to epilogue -> threshold -> top-k buffer.  No column windows, no probing loops.
//       GENERIC rows: accumulator tile of T 64-bit {column, partial sum} slots, direct-indexed when
//       the column window is <= T wide, otherwise open addressing (one ds_cmpst_rtn_b64 claims a
//       slot and deposits the first product); rows whose candidates do not fit are processed in
//       several column windows, exactly the reference's blocked path (s_plus.h:350-410), top-k state
//       carried across windows.  A sparse row that overflows its collision set falls back to this.
//     (measured on gfx950, scripts/lds_atomics_bench.hip: ds_add_f32 retires 0.33 lanes/clk/CU whatever
//      the address pattern, ds_cmpst_rtn_b64 3.3, ds_or/add_rtn_u32 ~10 — hence no float atomics on
//      the common path);
//   * candidates are pruned before any gather of the column terms Y*[c] by an upper bound of the
//     epilogue computed from per-launch minima of the Y vectors;
//   * the std::push_heap/pop_heap TopK becomes a workgroup-wide selection: survivors of the running
//     k-th value are appended to an LDS buffer and, when it fills, an MSD radix-select (4 x 8 bit over
//     an order-preserving key) keeps exactly k.
// HBM-bound integer/float streaming work: no MFMA on purpose.
//
// Everything below is written for gfx950 only (wave64, 160 KiB LDS).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "../../include/sp_knn.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int EMPTY = -1;                        // key of a free slot (column ids are >= 0)
constexpr u64 EMPTY64 = 0xFFFFFFFF00000000ull;   // free accumulator slot: key EMPTY, partial sum +0.0f
constexpr int MAX_PROBE = 128;   // probe budget of one element before a hashed window is declared overflowed
// m2 elements per lane and trip (two trips are in flight); 1024-thread workgroups have half the VGPR budget
#define ACC_UNROLL (NT >= 1024 ? 2 : NT >= 768 ? 4 : 8)
constexpr int U_SLACK = 1024;    // candidate-buffer entries beyond k (room between two selections)
// table slots per thread per drain iteration (their Y gathers fly together); 1024-thread workgroups have
// half the VGPR budget (128), where 8 would spill
#define DRAIN_UNROLL (NT >= 1024 ? 2 : NT >= 768 ? 4 : 8)

// scalar slots in LDS
enum { SH_CNT = 0, SH_OVF, SH_SEL, SH_NEED, SH_EQ, SH_CNT2, SH_RETRY, SH_QA, SH_QB, SH_DCTR, SH_PCTR, SH_NITEMS, SH_N };   // SH_CNT2/SEL/NEED/EQ belong to the selections

// phases timed by lane 0 of every workgroup when KParams::phase_cycles != NULL, then event counters
enum { PH_SETUP = 0, PH_SEGMENTS, PH_ACCUM, PH_DRAIN, PH_SELECT, PH_OUTPUT, PH_SWEEP1, PH_SWEEP2, PH_CSDRAIN,
       CT_ROWS_SPARSE, CT_ROWS_FALLBACK, CT_PASSES, PH_N };

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
    int cap;               // candidate buffer capacity (> k)
    u64 *gU;               // candidate buffers in global memory (only when they do not fit LDS)
    unsigned int *queue;   // [0] = next queue position (dynamic scheduling)
    const int4 *desc;      // [2*n_targets] row descriptors in queue order: {slot, m1 row, m1 start, m1 length}, {MACs (saturated), 0, 0, 0}
    unsigned m2_bytes;     // nnz(m2) * 4: extent of the m2 index / value buffers (buffer-load range check)
    int nb_log2;           // log2 of the sparse path's column bitmap size in bits (<= log2(T*64))
    int hash_fill;         // slots' worth of MACs one hash window may receive (= T * load_pct / 100)
    int static_sched;
    const float *ymin;     // [3] minima of Ytv / Ycos / Ydep over all columns (valid iff bound_ok)
    int bound_ok;          // weights/shrinks are all >= 0: the epilogue upper bound is sound
    int sparse_path;       // 1 = rows with few expected collisions take the bitmap path
    int fold;              // 1 = the single active column term (Ycos or Ydep) is already divided into m2_data: treat it as 1
    unsigned long long *phase_cycles;  // optional [PH_N]
    int dbg;               // ablation bits for profiling only (results are WRONG when non-zero):
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
        if (!epi.bound || epi.a1 != 1.f || !(t >= 0.f) || !(epi.bA > 0.f)) return;
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
template <int N>
__device__ __forceinline__ unsigned emit_candidates(const KParams &p, const RowCtx &rc, const int (&c)[N], const float (&xy)[N],
                                                    unsigned occ, u64 *U, int *sh) {
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
    unsigned want = 0;
    unsigned key[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float val = rc.epi(xy[j], ytv[j], ycos[j], ydep[j]);
        key[j] = fkey(val);
        if ((live & (1u << j)) && (val >= p.threshold) && (!rc.have_thr || key[j] > rc.thr_key)) want |= 1u << j;
    }
    // one aggregated reservation per wave
    unsigned stored = 0;
    wave_push<N>(want, &sh[SH_CNT], p.cap, &sh[SH_RETRY], [&](int j, int pos) {
        U[pos] = ((u64)key[j] << 32) | (u64)(unsigned)c[j];
        stored |= 1u << j;
    });
    return occ & (~want | stored);
}

// ---------------------------------------------------------------------------------------------
// helpers of the sparse path
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITEM = 256;         // m2 elements per work item: one 16-byte load per lane
constexpr int ITEM_CAP = 768;     // work items per row (LDS: 16 B each)
constexpr int POOL_BLK = 64;      // pool entries a wave reserves at a time (>= 64: one trip always fits a fresh block)
constexpr int CS_MAXPROBE = 64;   // linear-probe budget in the collision set
constexpr int SORT_MAX = 256;     // m1 rows up to this many entries are visited in descending |value| order

__device__ __forceinline__ int mbcnt64(u64 m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Wave-private window [pos, end) into a shared LDS pool: entries are appended with no atomic at all until
// the window is used up, then ONE returning atomic reserves the next POOL_BLK entries.  Abandoned tails stay
// zero ("hole"); consumers skip zeros.  pos/end are wave-uniform (scalar registers).
struct WavePool { int pos, end; };

template <typename W>
__device__ __forceinline__ void pool_push(WavePool &wp, bool pred, int *ctr, int cap, int *ovf, W &&write) {
    const u64 m = __ballot(pred);
    if (m == 0) return;                        // wave-uniform
    const int n = __popcll(m);
    if (wp.pos + n > wp.end) {
        int base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(ctr, POOL_BLK);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + POOL_BLK > cap) {           // pool exhausted: the row is redone on the generic path
            if ((threadIdx.x & 63) == 0) *ovf = 1;
            wp.pos = 0; wp.end = -1;
            return;
        }
        wp.pos = base; wp.end = base + POOL_BLK;
    }
    if (pred) write(wp.pos + mbcnt64(m));
    wp.pos += n;
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

// Selection for candidate buffers of at most 2*NT entries: every thread keeps its (<= 2) entries in registers,
// one LDS histogram per radix pass (hist4 = 4 x 256 counters, zero on entry and on exit), every wave scans the
// histogram redundantly (no broadcast barrier), 1 barrier per pass.
//   exact:  keeps exactly k entries; returns the key of the k-th largest.
//   !exact: stops after two passes (sign, exponent, 7 mantissa bits) when that already removes most of the
//           surplus: keeps every entry >= the lower edge of the 16-bit bin holding the k-th largest and returns
//           that edge — a valid (conservative) running cutoff, cheaper than the exact one.
// Must be entered by the whole workgroup.  Returns -1 when n <= k (nothing done).
template <int NT>
__device__ long long select_fast(u64 *U, int *hist4, int *sh, int k, bool exact) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = sh[SH_CNT];
    if (tid == 0) { sh[SH_CNT2] = 0; sh[SH_EQ] = 0; }   // (the generic path's selection leaves them dirty)
    __syncthreads();
    if (n <= k) return -1;
    u64 e[2];
    unsigned key[2];
    bool has[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + j * NT;
        has[j] = i < n;
        e[j] = has[j] ? U[i] : 0ull;
        key[j] = (unsigned)(e[j] >> 32);
    }
    unsigned prefix = 0;
    int need = k;
    int passes = 0;
    for (int ps = 0; ps < 4; ++ps) {
        const int shift = 24 - 8 * ps;
        const unsigned hmask = (ps == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
        int *h = hist4 + ps * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (has[j] && ((key[j] ^ prefix) & hmask) == 0u) atomicAdd(&h[(key[j] >> shift) & 255u], 1);
        __syncthreads();
        // lane L owns bins 255-4L .. 252-4L (lanes ascend as digits descend)
        const int4 c4 = *(const int4 *)&h[252 - 4 * lane];
        const int c0 = c4.w, c1 = c4.z, c2 = c4.y, c3 = c4.x;
        const int s = c0 + c1 + c2 + c3;
        const int incl = wave_incl_scan_dpp(s);
        const int excl = incl - s;
        const bool mine = excl < need && need <= incl;
        int d = 0, r = 0, cb = 0;
        if (mine) {
            const int b0 = 255 - 4 * lane;
            r = need - excl;
            if (r <= c0) { d = b0; cb = c0; }
            else if (r <= c0 + c1) { d = b0 - 1; r -= c0; cb = c1; }
            else if (r <= c0 + c1 + c2) { d = b0 - 2; r -= c0 + c1; cb = c2; }
            else { d = b0 - 3; r -= c0 + c1 + c2; cb = c3; }
        }
        const int leader = (int)__builtin_ctzll(__ballot(mine));
        d = __builtin_amdgcn_readlane(d, leader);
        r = __builtin_amdgcn_readlane(r, leader);
        cb = __builtin_amdgcn_readlane(cb, leader);
        const int above = need - r;            // entries of this pass's population that lie above the chosen bin
        prefix |= (unsigned)d << shift;
        passes = ps + 1;
        if (!exact && ps == 1) {
            // keeping the whole bin leaves (k - r) + cb entries: good enough when that is at most half the surplus
            const int kept = (k - r) + cb;
            (void)above;
            if (2 * (kept - k) <= (n - k)) { need = r; break; }
        }
        need = r;
    }
    const bool all_passes = (passes == 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
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
    __syncthreads();
    for (int i = tid; i < passes * 256; i += NT) hist4[i] = 0;
    if (tid == 0) sh[SH_CNT] = sh[SH_CNT2];
    __syncthreads();
    return (long long)prefix;
}

template <int NT, bool U_LDS>
__global__ __launch_bounds__(NT) void sp_knn_rows_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    constexpr int XB = (16 * NT + 256) > (ITEM_CAP * 16 + 4096) ? (16 * NT + 256) : (ITEM_CAP * 16 + 4096);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T;
    const int A_bytes = T * 8;

    // ---- LDS carve-up (single dynamic array) ----
    // region A  [0, T*8)      generic path: accumulator tile of T {column : partial dot} slots
    //                         sparse path : sweep 1: column bitmap (nb_bits);  sweep 2: [0,A/4) collision bitmap,
    //                                       [A/4,A/2) collision set, [A/2,A) survivor / member-product pool
    // region X  [.., +XB)     generic path: segment arrays;  sparse path: work items + 4 radix histograms
    // then hist / wsum / sh / ph, then the candidate buffer U (sparse sweep 1 borrows it for the duplicate pool)
    u64 *tab = (u64 *)smem;
    unsigned char *X = smem + A_bytes;
    int *seg_lo = (int *)X;                     // [NT]   start of the (window's) slice of m2 row u
    int *seg_pre = seg_lo + NT;                 // [NT+64] exclusive prefix of slice lengths
    float *seg_v1 = (float *)(seg_pre + NT + 64);  // [NT] m1 value of the segment
    int *seg_hi = (int *)(seg_v1 + NT);         // [NT]   end of the slice (= start of the next window's)
    int4 *items = (int4 *)X;                    // [ITEM_CAP] {m2 offset, count, m1 value bits, flat start}
    int *hist4 = (int *)(X + ITEM_CAP * 16);    // [4][256]
    int *hist = (int *)(X + XB);                // [256]
    int *wsum = hist + 256;                     // [64]
    int *sh = wsum + 64;                        // [32]
    u64 *ph = (u64 *)(sh + 32);                 // [16] phase timers / event counters (lane 0 only)
    u64 *U = U_LDS ? (u64 *)(ph + 16) : (p.gU + (size_t)blockIdx.x * (size_t)p.cap);

    // sparse path geometry
    const unsigned amask = (unsigned)((1u << (p.nb_log2 - 3)) - 1u) & ~3u;      // column -> byte of its bitmap word
    const int nb_bytes = 1 << (p.nb_log2 - 3);
    unsigned char *cbm = smem;
    const unsigned cmask = (unsigned)(A_bytes / 4 - 1) & ~3u;                   // column -> byte of its collision-bitmap word
    u64 *cs = (u64 *)(smem + A_bytes / 4);
    const int CSN = A_bytes / 32;
    const int cs_shift = 32 - (p.logT - 2);                                     // log2(CSN) = logT + 3 - 5
    u64 *pool = (u64 *)(smem + A_bytes / 2);
    const int pcap = A_bytes / 16;
    unsigned *dpool = (unsigned *)U;
    const int dcap = 2 * p.cap;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_indices, 0, (int)p.m2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_val = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_data, 0, (int)p.m2_bytes, 0x00020000);

    for (int i = tid; i < T; i += NT) tab[i] = EMPTY64;
    int lds_mode = 0;  // 0: region A is generic-clean (all EMPTY64), 1: sparse-clean (all zero, hist4 zero)
    if (tid < 32) sh[tid] = 0;
    if (tid < 16) ph[tid] = 0;
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    float ymin_tv = 0.f, ymin_cos = 0.f, ymin_dep = 0.f;
    if (p.bound_ok) {
        if (p.fold) { ymin_cos = 1.f; ymin_dep = 1.f; }     // folded column term: exactly 1 for every column
        else { ymin_tv = p.ymin[0]; ymin_cos = p.ymin[1]; ymin_dep = p.ymin[2]; }
    }

    // phase timers (lane 0 only; s_memtime ticks are shader cycles)
    const bool timing = (p.phase_cycles != nullptr) && tid == 0;
    u64 tmark = timing ? (u64)clock64() : 0;
#define PHASE_END(which) do { if (timing) { const u64 _n = (u64)clock64(); ph[which] += _n - tmark; tmark = _n; } } while (0)

    // Generic path streaming front end.  Visit the flat element space [eb, ee) of the current segment list
    // (nb segments, prefix in seg_pre): wave w owns a contiguous 64-aligned chunk, every lane handles AU
    // stride-64 elements per trip (coalesced loads), all lanes of a wave make the same number of trips, and the
    // loads of trip i+1 are issued before trip i is processed (two register sets, no copies).
    // Per lane the current segment is cached in registers (end of segment, flat->m2 index delta, m1 value):
    // the common element costs one compare and one add.  m2 is addressed with 32-bit byte offsets from the
    // scalar base pointers (the host only launches this kernel for nnz(m2) < 2^30).
    // body(c[], x[], v1[], valid): c = column id, x = m2 value (0 unless loadx), v1 = m1 value of the
    // element's segment; padding elements (bit clear in `valid`) repeat a real element of the lane, v1 = 0.
    const char *m2i_bytes = (const char *)p.m2_indices;
    const char *m2d_bytes = (const char *)p.m2_data;
    auto for_elements = [&](auto loadx, auto unroll, int eb, int ee, int nb, auto &&body) __attribute__((always_inline)) {
        constexpr bool LOADX = decltype(loadx)::value;
        constexpr int AU = decltype(unroll)::value;
        const int span = ee - eb;
        if (span <= 0) return;
        const int chunk = ((span + NW * 64 - 1) / (NW * 64)) * 64;
        const int e0 = eb + wave * chunk;
        const int e1 = min(e0 + chunk, ee);
        if (e0 >= e1) return;  // wave-uniform
        const int efirst = min(e0 + lane, e1 - 1);
        int sl = 0, sr = nb;  // last s in [0,nb) with seg_pre[s] <= efirst (seg_pre[0] = 0)
        while (sr - sl > 1) {
            const int mid = (sl + sr) >> 1;
            if (seg_pre[mid] <= efirst) sl = mid; else sr = mid;
        }
        int seg = sl;
        int seg_end = (seg + 1 < nb) ? seg_pre[seg + 1] : 0x7FFFFFFF;   // first flat index beyond the segment
        int delta = seg_lo[seg] - seg_pre[seg];                          // m2 position = flat index + delta
        float segv = seg_v1[seg];
        const int idx_safe = efirst + delta;
        auto fetch = [&](int ebase, int (&c)[AU], float (&x)[AU], float (&v1)[AU], unsigned &valid) {
            unsigned off[AU];
            valid = 0;
#pragma unroll
            for (int j = 0; j < AU; ++j) {
                const int ej = ebase + 64 * j + lane;
                const bool ok = ej < e1;
                if (ok && ej >= seg_end) {                 // rare: crossed into a later segment (skips empty ones)
                    do {
                        ++seg;
                        seg_end = (seg + 1 < nb) ? seg_pre[seg + 1] : 0x7FFFFFFF;
                    } while (ej >= seg_end);
                    delta = seg_lo[seg] - seg_pre[seg];
                    segv = seg_v1[seg];
                }
                off[j] = (unsigned)(ok ? ej + delta : idx_safe) << 2;
                v1[j] = ok ? segv : 0.f;
                valid |= ok ? (1u << j) : 0u;
            }
#pragma unroll
            for (int j = 0; j < AU; ++j) c[j] = *(const int *)(m2i_bytes + off[j]);
#pragma unroll
            for (int j = 0; j < AU; ++j) x[j] = LOADX ? *(const float *)(m2d_bytes + off[j]) : 0.f;
        };
        constexpr int STEP = 64 * AU;
        int cA[AU], cB[AU];
        float xA[AU], xB[AU], vA[AU], vB[AU];
        unsigned validA = 0, validB = 0;
        fetch(e0, cA, xA, vA, validA);
        for (int ebase = e0; ebase < e1; ebase += 2 * STEP) {
            const bool hasB = ebase + STEP < e1;          // wave-uniform
            if (hasB) fetch(ebase + STEP, cB, xB, vB, validB);
            body(cA, xA, vA, validA);
            if (hasB) {
                if (ebase + 2 * STEP < e1) fetch(ebase + 2 * STEP, cA, xA, vA, validA);
                body(cB, xB, vB, validB);
            }
        }
    };

    // Turn per-thread slice lengths into the flat prefix array; returns the total.  Two barriers.
    auto scan_segments = [&](int len) -> int {
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
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(total);
    };

    // ---- row pipeline ----
    // The chain  queue -> descriptor {slot, row, m1 start, m1 length, MACs} -> m1 entries -> m2 row bounds  is four
    // dependent global loads (~1 us each under load).  It is software-pipelined across rows: while row r is
    // processed, the queue slot of row r+3 is claimed, the descriptor of row r+2 is loaded, the m1 entries of
    // row r+1 are loaded (top of the row) and its m2 row bounds fetched (middle of the row).
    const int4 *desc = p.desc;
    auto load_desc = [&](int q, int4 &d0, int &work) {
        d0 = make_int4(-1, 0, 0, 0);
        work = 0;
        if (q < p.n_targets) { d0 = desc[2 * (size_t)q]; work = desc[2 * (size_t)q + 1].x; }
    };
    int q_nn = 0;      // queue index two rows ahead (static schedule: computed; dynamic: through LDS)
    int pend_q = 0;    // (tid 0) claimed queue index three rows ahead
    int4 dC, dN;       // descriptors of the current and the next row
    int wC = 0, wN = 0;
    if (p.static_sched) {
        load_desc((int)blockIdx.x, dC, wC);
        load_desc((int)(blockIdx.x + gridDim.x), dN, wN);
        q_nn = (int)(blockIdx.x + 2 * gridDim.x);
    } else {
        if (tid == 0) {
            sh[SH_QA] = (int)atomicAdd(&p.queue[0], 1u);
            sh[SH_QB] = (int)atomicAdd(&p.queue[0], 1u);
            pend_q = (int)atomicAdd(&p.queue[0], 1u);
        }
        __syncthreads();
        load_desc(sh[SH_QA], dC, wC);
        load_desc(sh[SH_QB], dN, wN);
        __syncthreads();
    }
    // m1 entries / m2 row bounds held per thread (segment `tid` of the row) for rows with at most NT entries
    int my_r0 = 0, my_len = 0;
    float my_v = 0.f;
    if (dC.x >= 0 && tid < dC.w && dC.w <= NT) {
        const int u = p.m1_indices[dC.z + tid];
        my_v = p.m1_data[dC.z + tid];
        my_r0 = p.m2_indptr[u];
        my_len = p.m2_indptr[u + 1] - my_r0;
    }

    for (;;) {
        // row-constant values are wave-uniform: v_readfirstlane moves them to scalar registers, which frees
        // vector registers for the streaming loops
        const int slot_i = __builtin_amdgcn_readfirstlane(dC.x);
        if (slot_i < 0) break;
        const int t = __builtin_amdgcn_readfirstlane(dC.y);
        const int s1 = __builtin_amdgcn_readfirstlane(dC.z);
        const int n1 = __builtin_amdgcn_readfirstlane(dC.w);
        const unsigned macs32 = (unsigned)__builtin_amdgcn_readfirstlane(wC);
        const u64 macs = (u64)macs32;   // saturated at 2^32-1 by the work prepass

        // prefetch: queue slot three rows ahead, m1 entries of the next row
        if (!p.static_sched && tid == 0) {
            sh[SH_QA] = pend_q;
            pend_q = (int)atomicAdd(&p.queue[0], 1u);
        }
        const bool nx_regs = dN.x >= 0 && dN.w <= NT;
        int nx_u = 0;
        float nx_v = 0.f;
        if (nx_regs && tid < dN.w) {
            nx_u = p.m1_indices[dN.z + tid];
            nx_v = p.m1_data[dN.z + tid];
        }
        int nx_r0 = 0, nx_len = 0;
        bool b2_done = false;
        auto fetch_next_bounds = [&]() __attribute__((always_inline)) {
            if (!b2_done) {
                if (nx_regs && tid < dN.w) {
                    nx_r0 = p.m2_indptr[nx_u];
                    nx_len = p.m2_indptr[nx_u + 1] - nx_r0;
                }
                b2_done = true;
            }
        };
        __syncthreads();
        int4 dNN;
        int wNN;
        if (!p.static_sched) q_nn = sh[SH_QA];
        load_desc(q_nn, dNN, wNN);
        if (p.static_sched) q_nn += (int)gridDim.x;

        RowCtx rc;
        rc.row = t;
        rc.have_thr = false;
        rc.thr_key = 0;
        Epi &epi = rc.epi;
        epi.a1 = p.a1; epi.l1 = p.l1; epi.l2 = p.l2; epi.l3 = p.l3; epi.t1 = p.t1; epi.t2 = p.t2;
        epi.stab = p.stab; epi.bayes = p.bayes; epi.threshold = p.threshold; epi.any = any_norm;
        auto rflf = [](float v) __attribute__((always_inline)) {
            return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
        };
        epi.xtv = (p.l1 != 0.f) ? rflf(p.Xtv[t]) : 0.f;
        epi.xcos = (p.l2 != 0.f) ? rflf(p.Xcos[t]) : 0.f;
        epi.xdep = (p.l3 != 0.f) ? rflf(p.Xdep[t]) : 0.f;
        // den = l1*(t1*(X-xy) + t2*(Y-xy) + xy) + l2*Xc*Yc + l3*Xd*Yd + stab  >=  bA + bB*xy  when the
        // column terms are replaced by their minima and their multipliers are non-negative
        epi.bound = p.bound_ok && !(epi.xcos < 0.f) && !(epi.xdep < 0.f);
        epi.bA = p.l1 * (p.t1 * epi.xtv + p.t2 * ymin_tv) + p.l2 * epi.xcos * ymin_cos + p.l3 * epi.xdep * ymin_dep + p.stab;
        epi.bB = p.l1 * (1.f - p.t1 - p.t2);

        rc.set_cut(p.threshold);
        rc.f0 = rc.f1 = rc.g0 = rc.g1 = 0;
        if (p.filter_mode == SP_SEL_MATRIX) { rc.f0 = __builtin_amdgcn_readfirstlane(p.f_indptr[t]); rc.f1 = __builtin_amdgcn_readfirstlane(p.f_indptr[t + 1]); }
        if (p.target_mode == SP_SEL_MATRIX) { rc.g0 = __builtin_amdgcn_readfirstlane(p.t_indptr[t]); rc.g1 = __builtin_amdgcn_readfirstlane(p.t_indptr[t + 1]); }

        // running k-th value after a selection
        auto took_threshold = [&](long long thr_new) __attribute__((always_inline)) {
            if (thr_new >= 0) {
                rc.have_thr = true;
                rc.thr_key = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)thr_new);
                rc.set_cut(p.threshold);
            }
        };
        // selection on the sparse path (hist4 lives in region X, which the generic path uses for its segments)
        auto select_sparse = [&](bool exact) __attribute__((always_inline)) {
            if (p.cap <= 2 * NT) took_threshold(select_fast<NT>(U, hist4, sh, p.k, exact));
            else took_threshold(compact_topk<NT>(U, hist, sh, p.k));
        };

        // =========================================================================================
        // SPARSE path: few products share a column (the KNN / recommender shape)
        //   sweep 1 (column ids): one bit per column in an LDS bitmap (exact while n_cols <= bits); a product
        //     that finds its bit set has its column appended to a duplicate pool;
        //   the bitmap is cleared, the duplicate columns become a small collision set + a collision bitmap;
        //   sweep 2 (ids + values): ONE bit test per product — columns of the collision set have their product
        //     appended to a pool (accumulated densely afterwards), every other product is the only one of its
        //     column and is appended only if its raw dot can still beat the running k-th value;
        //   pools are consumed by dense phases: epilogue -> threshold -> top-k buffer -> selection.
        // Work is handed out in items of <= 256 consecutive elements of one m2 row: the row base, the count
        // and the m1 value are scalars, one 16-byte buffer load per lane fetches a whole item.
        // =========================================================================================
        bool row_done = (macs == 0);
        bool sparse_ok = false;
        if (!row_done && p.sparse_path && n1 <= SORT_MAX && n1 <= NT && p.n_cols > T && macs32 < (1u << 30)) {
            const float m = (float)macs32;
            const float alias = (p.nb_log2 < 31 && (1 << p.nb_log2) < p.n_cols) ? 1.f / (float)(1 << p.nb_log2) : 0.f;
            const float expect = 0.5f * m * m * (1.f / (float)p.n_cols + alias);
            sparse_ok = expect <= 0.30f * (float)CSN && expect <= 0.40f * (float)dcap;
        }
        int n_items = 0, my_ib = 0, my_fs = 0;
        if (sparse_ok) {
            if (lds_mode != 1) {
                for (int i = tid; i < T / 2; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
                for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
                lds_mode = 1;
            }
            // the duplicate pool borrows U's storage (empty until sweep 2); holes must read zero
            for (int i = tid; i < p.cap / 2; i += NT) ((int4 *)U)[i] = make_int4(0, 0, 0, 0);
            if (tid == 0) { sh[SH_DCTR] = 0; sh[SH_PCTR] = 0; sh[SH_NITEMS] = 0; }
            // Segments are visited in descending |m1 value| order: each segment scales its m2 row by its own m1
            // value, so the heavy segments first make the running k-th value rise early and the survivor rate
            // fall monotonically.  Rank, first item and flat start of every segment come from one all-pairs pass
            // spread over the whole workgroup (n1 <= 256).
            int *keyS = (int *)items, *lenS = keyS + SORT_MAX, *ibS = lenS + SORT_MAX, *fsS = ibS + SORT_MAX;
            if (tid < SORT_MAX) { ibS[tid] = 0; fsS[tid] = 0; }
            if (tid < n1) { keyS[tid] = (int)(__float_as_uint(my_v) & 0x7FFFFFFFu); lenS[tid] = my_len; }
            __syncthreads();
            {
                const int n1p = (n1 + 63) & ~63;
                const int parts = NT / n1p;
                const int seg = tid % n1p, part = tid / n1p;
                if (seg < n1 && part < parts) {
                    const int key = keyS[seg];
                    int ib = 0, fs = 0;
                    for (int j = part; j < n1; j += parts) {
                        const int kj = keyS[j], lj = lenS[j];     // same address across the wave: broadcast reads
                        const bool before = (kj > key) || (kj == key && j < seg);
                        ib += before ? (lj + ITEM - 1) / ITEM : 0;
                        fs += before ? lj : 0;
                    }
                    if (ib) atomicAdd(&ibS[seg], ib);
                    if (fs) atomicAdd(&fsS[seg], fs);
                }
                if (tid < n1 && my_len > 0) atomicAdd(&sh[SH_NITEMS], (my_len + ITEM - 1) / ITEM);
            }
            __syncthreads();
            if (tid < n1) { my_ib = ibS[tid]; my_fs = fsS[tid]; }
            n_items = sh[SH_NITEMS];
            __syncthreads();                    // scratch read before the items overwrite it
            if (n_items > ITEM_CAP) sparse_ok = false;
        }
        if (sparse_ok) {
            if (tid < n1) {
                int q = 0;
                for (int o = 0; o < my_len; o += ITEM, ++q)
                    items[my_ib + q] = make_int4(my_r0 + o, min(ITEM, my_len - o), (int)__float_as_uint(my_v), my_fs + o);
            }
            __syncthreads();
            PHASE_END(PH_SETUP);

            // ---- sweep 1: column ids only ----
            {
                WavePool wp{0, -1};
                auto ld = [&](int it, unsigned (&c)[4], int &cnt) __attribute__((always_inline)) {
                    const int4 d = items[it];
                    const int off = __builtin_amdgcn_readfirstlane(d.x);
                    cnt = __builtin_amdgcn_readfirstlane(d.y);
                    if (cnt == ITEM) {
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, lane * 16, off * 4, 0);
                        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_idx, (j * 64 + lane) * 4, off * 4, 0);
                    }
                };
                auto body = [&](const unsigned (&c)[4], int cnt) __attribute__((always_inline)) {
                    unsigned old[4], bit[4];
                    if (cnt == ITEM) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            bit[j] = 1u << (c[j] & 31u);
                            old[j] = atomicOr((unsigned *)(smem + ((c[j] >> 3) & amask)), bit[j]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            bit[j] = (j * 64 + lane < cnt) ? (1u << (c[j] & 31u)) : 0u;     // padding ORs nothing
                            old[j] = atomicOr((unsigned *)(smem + ((c[j] >> 3) & amask)), bit[j]);
                        }
                    }
                    bool dup[4];
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { dup[j] = (old[j] & bit[j]) != 0u; any |= dup[j]; }
                    if (__ballot(any)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            pool_push(wp, dup[j], &sh[SH_DCTR], dcap, &sh[SH_OVF], [&](int pos) { dpool[pos] = ~c[j]; });
                    }
                };
                unsigned cA[4], cB[4];
                int nA = 0, nB = 0;
                int it = wave;
                if (it < n_items) ld(it, cA, nA);
                while (it < n_items) {
                    const int it2 = it + NW;
                    if (it2 < n_items) ld(it2, cB, nB);
                    body(cA, nA);
                    if (it2 >= n_items) break;
                    const int it3 = it2 + NW;
                    if (it3 < n_items) ld(it3, cA, nA);
                    body(cB, nB);
                    it = it3;
                }
            }
            __syncthreads();
            const int ovf1 = sh[SH_OVF];
            const int dext = min(sh[SH_DCTR], dcap);
            fetch_next_bounds();
            // the bitmap has done its job: back to zero (16-byte stores), then the collision structures go there
            for (int i = tid; i < (nb_bytes >> 4); i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
            __syncthreads();
            bool failed = (ovf1 != 0);
            if (!failed) {
                for (int i = tid; i < dext; i += NT) {
                    const unsigned nc = dpool[i];
                    if (nc != 0u) {
                        const unsigned c = ~nc;
                        unsigned h = hash_bits((int)c, 2654435761u, cs_shift);
                        int tries = 0;
                        for (; tries < CS_MAXPROBE; ++tries) {
                            const u64 prev = atomicCAS(&cs[h], 0ull, (u64)nc << 32);     // {~column : +0.0f}
                            if (prev == 0ull) { atomicOr((unsigned *)(cbm + ((c >> 3) & cmask)), 1u << (c & 31u)); break; }
                            if ((unsigned)(prev >> 32) == nc) break;                     // already a member
                            h = (h + 1u) & (unsigned)(CSN - 1);
                        }
                        if (tries == CS_MAXPROBE) sh[SH_OVF] = 1;
                    }
                }
                __syncthreads();
                failed = (sh[SH_OVF] != 0);
            }
            PHASE_END(PH_SWEEP1);

            if (!failed) {
                // ---- sweep 2 over items [i0, i1) ----
                auto sweep2 = [&](int i0, int i1) __attribute__((always_inline)) {
                    WavePool wp{0, -1};
                    auto ld = [&](int it, unsigned (&c)[4], float (&v)[4], int &cnt, float &segv) __attribute__((always_inline)) {
                        const int4 d = items[it];
                        const int off = __builtin_amdgcn_readfirstlane(d.x);
                        cnt = __builtin_amdgcn_readfirstlane(d.y);
                        segv = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(d.z));
                        if (cnt == ITEM) {
                            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, lane * 16, off * 4, 0);
                            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs_val, lane * 16, off * 4, 0);
                            c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
                            v[0] = __uint_as_float(b.x); v[1] = __uint_as_float(b.y); v[2] = __uint_as_float(b.z); v[3] = __uint_as_float(b.w);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_idx, (j * 64 + lane) * 4, off * 4, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_val, (j * 64 + lane) * 4, off * 4, 0));
                        }
                    };
                    auto body = [&](const unsigned (&c)[4], const float (&v)[4], int cnt, float segv) __attribute__((always_inline)) {
                        unsigned w[4];
                        float x[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            x[j] = v[j] * segv;
                            w[j] = *(const unsigned *)(cbm + ((c[j] >> 3) & cmask));
                        }
                        bool mem[4], push[4];
                        bool any = false;
                        const bool full = (cnt == ITEM);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = full || (j * 64 + lane < cnt);
                            mem[j] = ok && (((w[j] >> (c[j] & 31u)) & 1u) != 0u);
                            // a product outside the collision set is the only one of its column: keep it only if
                            // its raw dot can still enter the top-k (NaN stays: the exact judge drops it)
                            push[j] = mem[j] || (ok && !(x[j] <= rc.xy_cut));
                            any |= push[j];
                        }
                        if (__ballot(any)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                pool_push(wp, push[j], &sh[SH_PCTR], pcap, &sh[SH_OVF], [&](int pos) {
                                    pool[pos] = ((u64)((c[j] + 1u) | (mem[j] ? 0x80000000u : 0u)) << 32) | (u64)__float_as_uint(x[j]);
                                });
                        }
                    };
                    unsigned cA[4], cB[4];
                    float vA[4], vB[4];
                    int nA = 0, nB = 0;
                    float sA = 0.f, sB = 0.f;
                    int it = i0 + wave;
                    if (it < i1) ld(it, cA, vA, nA, sA);
                    while (it < i1) {
                        const int it2 = it + NW;
                        if (it2 < i1) ld(it2, cB, vB, nB, sB);
                        body(cA, vA, nA, sA);
                        if (it2 >= i1) break;
                        const int it3 = it2 + NW;
                        if (it3 < i1) ld(it3, cA, vA, nA, sA);
                        body(cB, vB, nB, sB);
                        it = it3;
                    }
                };
                // ---- dense consumer of the pool: member products accumulate in the collision set, single
                // products are judged (column terms, epilogue, threshold) and appended to U; a full U triggers a
                // selection and another pass over what is left.  Leaves the pool all zero. ----
                auto consume_pool = [&]() __attribute__((always_inline)) {
                    const int ext = min(sh[SH_PCTR], pcap);
                    for (;;) {
                        for (int base = 0; base < ext; base += NT * DRAIN_UNROLL) {
                            u64 e[DRAIN_UNROLL];
                            int c[DRAIN_UNROLL];
                            float xy[DRAIN_UNROLL];
                            unsigned occ = 0;
#pragma unroll
                            for (int j = 0; j < DRAIN_UNROLL; ++j) {
                                const int idx = base + j * NT + tid;
                                e[j] = (idx < ext) ? pool[idx] : 0ull;
                            }
#pragma unroll
                            for (int j = 0; j < DRAIN_UNROLL; ++j) {
                                const unsigned hi = (unsigned)(e[j] >> 32);
                                c[j] = (int)((hi & 0x7FFFFFFFu) - 1u);
                                xy[j] = __uint_as_float((unsigned)e[j]);
                                if (e[j] != 0ull) {
                                    bool single = (hi >> 31) == 0u;
                                    if (!single) {
                                        const unsigned nc = ~(unsigned)c[j];
                                        unsigned h = hash_bits(c[j], 2654435761u, cs_shift);
                                        single = true;            // bit aliasing: flagged but not in the set
                                        for (int tries = 0; tries < CS_MAXPROBE; ++tries) {
                                            const u64 s = cs[h];
                                            if ((unsigned)(s >> 32) == nc) { atomicAdd((float *)&cs[h], xy[j]); single = false; break; }
                                            if (s == 0ull) break;
                                            h = (h + 1u) & (unsigned)(CSN - 1);
                                        }
                                    }
                                    if (single && !(xy[j] <= rc.xy_cut)) occ |= 1u << j;
                                }
                            }
                            const unsigned done = emit_candidates<DRAIN_UNROLL>(p, rc, c, xy, occ, U, sh);
#pragma unroll
                            for (int j = 0; j < DRAIN_UNROLL; ++j)
                                if (e[j] != 0ull && (!(occ & (1u << j)) || (done & (1u << j)))) pool[base + j * NT + tid] = 0ull;
                        }
                        __syncthreads();
                        const int retry = sh[SH_RETRY];
                        if (!retry) break;  // uniform
                        __syncthreads();
                        if (tid == 0) {
                            sh[SH_RETRY] = 0;
                            if (sh[SH_CNT] > p.cap) sh[SH_CNT] = p.cap;   // failed appends over-counted
                        }
                        __syncthreads();
                        PHASE_END(PH_DRAIN);
                        select_sparse(false);
                        PHASE_END(PH_SELECT);
                    }
                    if (tid == 0) sh[SH_PCTR] = 0;
                };

                // The products are offered in growing chunks with a selection after each: the first chunk is
                // small enough that accepting everything cannot overflow U; once the k-th best of n products is
                // known, about k*m/n of the next m would survive in an exchangeable stream — far fewer here,
                // because segments come in descending weight — so the next chunk may be 4*n*(cap-k)/k long.
                const int room = p.cap - min(p.k, p.cap - 1);
                int i0 = 0;
                long long chunk = room;
                while (i0 < n_items) {
                    const int i1 = (int)min((long long)n_items, (long long)i0 + max(1ll, chunk / ITEM));
                    sweep2(i0, i1);
                    __syncthreads();
                    if (sh[SH_OVF]) { failed = true; break; }     // pool overflowed: dropped products cannot be re-offered
                    PHASE_END(PH_SWEEP2);
                    consume_pool();
                    i0 = i1;
                    const long long pos = (i0 < n_items) ? (long long)items[i0].w : (long long)macs32;
                    const int n_now = sh[SH_CNT];
                    __syncthreads();                               // SH_PCTR reset / SH_CNT read before anything moves on
                    PHASE_END(PH_DRAIN);
                    if (i0 < n_items && n_now > p.k) {
                        select_sparse(false);
                        PHASE_END(PH_SELECT);
                    }
                    chunk = rc.have_thr ? max((long long)room, 4ll * pos * (long long)room / (long long)p.k) : (long long)room;
                }
            }

            if (!failed) {
                // ---- drain the collision set (complete sums now), clearing it and its bitmap bits ----
                for (;;) {
                    for (int base = 0; base < CSN; base += NT * DRAIN_UNROLL) {
                        int c[DRAIN_UNROLL];
                        float xy[DRAIN_UNROLL];
                        unsigned occ = 0;
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) {
                            const int idx = base + j * NT + tid;
                            const u64 s = (idx < CSN) ? cs[idx] : 0ull;
                            c[j] = (int)~(unsigned)(s >> 32);
                            xy[j] = __uint_as_float((unsigned)s);
                            if (s != 0ull) occ |= 1u << j;
                        }
                        const unsigned done = emit_candidates<DRAIN_UNROLL>(p, rc, c, xy, occ, U, sh);
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) {
                            if (done & (1u << j)) {
                                cs[base + j * NT + tid] = 0ull;
                                atomicAnd((unsigned *)(cbm + (((unsigned)c[j] >> 3) & cmask)), ~(1u << ((unsigned)c[j] & 31u)));
                            }
                        }
                    }
                    __syncthreads();
                    const int retry = sh[SH_RETRY];
                    if (!retry) break;  // uniform
                    __syncthreads();
                    if (tid == 0) {
                        sh[SH_RETRY] = 0;
                        if (sh[SH_CNT] > p.cap) sh[SH_CNT] = p.cap;
                    }
                    __syncthreads();
                    PHASE_END(PH_CSDRAIN);
                    select_sparse(false);
                    PHASE_END(PH_SELECT);
                }
                PHASE_END(PH_CSDRAIN);
                row_done = true;
                if (timing) ph[CT_ROWS_SPARSE] += 1;
            } else {
                // a pool or the collision set overflowed: forget this attempt, take the generic path
                for (int i = tid; i < T; i += NT) tab[i] = EMPTY64;
                lds_mode = 0;
                if (tid == 0) { sh[SH_CNT] = 0; sh[SH_OVF] = 0; sh[SH_RETRY] = 0; sh[SH_CNT2] = 0; sh[SH_EQ] = 0; }
                rc.have_thr = false;
                rc.thr_key = 0;
                rc.set_cut(p.threshold);
                __syncthreads();
                if (timing) ph[CT_ROWS_FALLBACK] += 1;
            }
        }

        // =========================================================================================
        // GENERIC path: accumulator tile + column windows
        // =========================================================================================
        if (!row_done) {
            if (lds_mode != 0) {
                for (int i = tid; i < T; i += NT) tab[i] = EMPTY64;
                lds_mode = 0;
                __syncthreads();
            }
            bool retry_window = false;  // the current window repeats the previous lo (after an overflow)

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
                    const int total = scan_segments(len);
                    PHASE_END(PH_SEGMENTS);

                    for_elements(std::true_type{}, std::integral_constant<int, ACC_UNROLL>{}, 0, total, nb,
                                 [&](const int (&c)[ACC_UNROLL], const float (&xr)[ACC_UNROLL], const float (&v1)[ACC_UNROLL], unsigned) {
                        float x[ACC_UNROLL];
#pragma unroll
                        for (int j = 0; j < ACC_UNROLL; ++j) x[j] = xr[j] * v1[j];   // padding elements carry 0
                        if (p.dbg & 1) {
                            float sink = 0.f;
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) sink += x[j] + (float)c[j];
                            if (sink == 123.456f) sh[SH_OVF] = 2;  // keeps the loads alive, never true in practice
                        } else if (dense) {
                            // direct-indexed window: every column owns its slot.  Optimistic update: read the
                            // slot, then ONE 64-bit compare-and-swap writes {column, sum + x} (ds_cmpst_rtn_b64:
                            // 3.3 lanes/clk against 0.33 for ds_add_f32); the lanes of a wave instruction hold 64
                            // distinct columns of one m2 row, so only another wave can interfere — a lost race
                            // falls back to the hardware float add on the sum half (the key half is already set
                            // by whoever won), which cannot livelock on hot columns.
                            u64 cur[ACC_UNROLL], prev[ACC_UNROLL];
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) cur[j] = tab[c[j] - wlo];
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) {
                                const float sum = __uint_as_float((unsigned)cur[j]) + x[j];
                                prev[j] = atomicCAS(&tab[c[j] - wlo], cur[j], ((u64)(unsigned)c[j] << 32) | (u64)__float_as_uint(sum));
                            }
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j)
                                if (prev[j] != cur[j]) atomicAdd((float *)&tab[c[j] - wlo], x[j]);
                        } else {
                            // Hashed window.  One 64-bit compare-and-swap claims a free slot for a new column AND
                            // deposits its first product; finding the same column already there turns into a
                            // hardware float add on the sum half (slow on gfx950, 3 clk/lane, but immune to
                            // contention on hot columns); finding another column means double-hash probing.
                            // Round 1 issues the ACC_UNROLL claims back to back; the few leftovers are then walked
                            // one element per lane per round.
                            unsigned hs[ACC_UNROLL];
                            u64 prev[ACC_UNROLL];
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) hs[j] = ((unsigned)c[j] * 2654435761u) >> hshift;
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j)
                                prev[j] = atomicCAS(&tab[hs[j]], EMPTY64, ((u64)(unsigned)c[j] << 32) | (u64)__float_as_uint(x[j]));
                            unsigned pend = 0;
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) {
                                const bool hit = ((int)(prev[j] >> 32) == c[j]);
                                if (hit) atomicAdd((float *)&tab[hs[j]], x[j]);
                                if (prev[j] != EMPTY64 && !hit) pend |= 1u << j;
                            }
                            int plen = 0;
                            while (__ballot(pend != 0)) {   // wave-uniform trip count
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
                                else if (++plen >= MAX_PROBE) { sh[SH_OVF] = 1; pend = 0; }
#pragma unroll
                                for (int j = 0; j < ACC_UNROLL; ++j)
                                    if (bit == (1u << j)) hs[j] = hh;
                            }
                        }
                    });
                    __syncthreads();  // seg_* are rewritten by the next batch
                    PHASE_END(PH_ACCUM);
                }

                // ================= overflow: discard the window, halve it, retry =================
                if (!dense) {
                    const int ovf = sh[SH_OVF];
                    __syncthreads();
                    if (ovf) {
                        for (int i = tid; i < t_eff; i += NT) tab[i] = EMPTY64;
                        if (tid == 0) sh[SH_OVF] = 0;
                        width = max((long long)T, (width + 1) / 2);
                        retry_window = true;  // same lo again: slice starts are still in seg_lo
                        __syncthreads();
                        continue;
                    }
                }
                retry_window = false;
                if (timing) ph[CT_PASSES] += 1;

                // ================= drain: one barrier-free sweep, overflow-retry =================
                for (;;) {
                    for (int base = 0; base < t_eff; base += NT * DRAIN_UNROLL) {
                        int c[DRAIN_UNROLL];
                        float xy[DRAIN_UNROLL];
                        unsigned occ = 0;
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
                        const unsigned done = emit_candidates<DRAIN_UNROLL>(p, rc, c, xy, occ, U, sh);
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j)
                            if (done & (1u << j)) tab[base + j * NT + tid] = EMPTY64;
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
                    took_threshold(compact_topk<NT>(U, hist, sh, p.k));
                    PHASE_END(PH_SELECT);
                }
                PHASE_END(PH_DRAIN);
                lo = hi;
            }
        }

        // ================= final selection + write-out =================
        fetch_next_bounds();
        __syncthreads();
        const int n_fin = sh[SH_CNT];
        __syncthreads();
        if (n_fin > p.k) {
            if (lds_mode == 1) select_sparse(true);
            else took_threshold(compact_topk<NT>(U, hist, sh, p.k));
        }
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
        if (tid == 0 && p.counts) p.counts[slot_i] = n_out;
        __syncthreads();
        if (tid == 0) sh[SH_CNT] = 0;
        // rotate the row pipeline
        dC = dN; wC = wN;
        dN = dNN; wN = wNN;
        my_r0 = nx_r0; my_len = nx_len; my_v = nx_v;
        __syncthreads();
        PHASE_END(PH_OUTPUT);
    }
    if (timing) {
#pragma unroll
        for (int i = 0; i < PH_N; ++i) atomicAdd(&p.phase_cycles[i], ph[i]);
    }
#undef PHASE_END
}

// ---- work-ordered row queue (longest-processing-time-first, to within a factor 2) ----
// Rows are visited in descending MACs(t) buckets (bucket = floor(log2(work))), so that one huge row at the end of
// the target list cannot become the tail of the launch on skewed (power-law) matrices.
__global__ __launch_bounds__(256) void sp_row_work_kernel(int n_targets, const int *targets, const int *m1_indices,
                                                           const int *m1_indptr, const int *m2_indptr, unsigned *work,
                                                           unsigned *bucket_count) {
    __shared__ unsigned hist[32];
    if (threadIdx.x < 32) hist[threadIdx.x] = 0;
    __syncthreads();
    const int gw = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (gw < n_targets) {
        const int t = targets[gw];
        const int s = m1_indptr[t], e = m1_indptr[t + 1];
        u64 acc = 0;
        for (int j = s + lane; j < e; j += 64) {
            const int u = m1_indices[j];
            acc += (u64)(m2_indptr[u + 1] - m2_indptr[u]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (lane == 0) {
            const unsigned w = acc > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)acc;
            work[gw] = w;
            atomicAdd(&hist[31 - __clz((int)(w | 1u))], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && hist[threadIdx.x]) atomicAdd(&bucket_count[threadIdx.x], hist[threadIdx.x]);
}

// bucket_count[0..32) -> bucket_base[0..32): start of each bucket when buckets are laid out heaviest first
// bucket_base[32] = 1 when the work spans at least a factor ~4 (otherwise the target order is kept: nothing to gain)
__global__ void sp_bucket_base_kernel(const unsigned *bucket_count, unsigned *bucket_base) {
    if (threadIdx.x == 0) {
        unsigned run = 0;
        int hi = -1, lo = 32;
        for (int b = 31; b >= 0; --b) {
            bucket_base[b] = run;
            run += bucket_count[b];
            if (bucket_count[b]) { if (hi < 0) hi = b; lo = b; }
        }
        bucket_base[32] = (hi - lo >= 2) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void sp_row_order_kernel(int n_targets, const unsigned *work, unsigned *bucket_base, int *order) {
    if (bucket_base[32] == 0) return;   // uniform work: the main kernel keeps the target order
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < n_targets;
    const int b = live ? 31 - __clz((int)(work[i] | 1u)) : -1;
    // one atomic per (wave, bucket): rows of similar work share a bucket, and a single global word only
    // sustains ~88 atomics/us — a per-row atomic would cost ~11 ms for 1M equal rows
    u64 todo = __ballot(live);
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const int b0 = __shfl(b, leader, 64);
        const u64 same = __ballot(live && b == b0);
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&bucket_base[b0], (unsigned)__popcll(same));
        base = __shfl(base, leader, 64);
        if (live && b == b0) order[base + __popcll(same & ((1ull << lane) - 1ull))] = i;
        todo &= ~same;
    }
}

// Row descriptors in queue order: what the main kernel needs to start a row, one 32-byte record per queue
// position, so that its dependent-load chain is queue -> descriptor -> m1 entries -> m2 row bounds.
__global__ __launch_bounds__(256) void sp_row_desc_kernel(int n_targets, const int *targets, const int *m1_indptr, const unsigned *work,
                                                           const unsigned *ordered_flag, const int *order, int4 *desc) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= n_targets) return;
    const int slot = (ordered_flag != nullptr && ordered_flag[0] != 0u) ? order[pos] : pos;
    const int t = targets[slot];
    const int s = m1_indptr[t], e = m1_indptr[t + 1];
    desc[2 * (size_t)pos] = make_int4(slot, t, s, e - s);
    desc[2 * (size_t)pos + 1] = make_int4((int)work[slot], 0, 0, 0);
}

// Minima of the three column-term vectors over all columns (one workgroup; feeds Epi::upper).
__global__ __launch_bounds__(1024) void sp_colterm_min_kernel(int n_cols, const float *Ytv, const float *Ycos, const float *Ydep, float *out) {
    __shared__ float red[3][16];
    const float inf = __builtin_inff();
    float m0 = inf, m1 = inf, m2 = inf;
    for (int i = threadIdx.x; i < n_cols; i += 1024) {
        if (Ytv) m0 = fminf(m0, Ytv[i]);
        if (Ycos) m1 = fminf(m1, Ycos[i]);
        if (Ydep) m2 = fminf(m2, Ydep[i]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        m0 = fminf(m0, __shfl_xor(m0, d, 64));
        m1 = fminf(m1, __shfl_xor(m1, d, 64));
        m2 = fminf(m2, __shfl_xor(m2, d, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m0; red[1][threadIdx.x >> 6] = m1; red[2][threadIdx.x >> 6] = m2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float m = inf;
        for (int w = 0; w < 16; ++w) m = fminf(m, red[threadIdx.x][w]);
        out[threadIdx.x] = (m == inf) ? 0.f : m;   // vector not in use (or empty): its weight is 0 anyway
    }
}

// Fold the column term of a product-form epilogue into the m2 stream:  out[i] = data[i] / Y[indices[i]]
// (0 where Y is 0: the reference returns 0 for a zero denominator, s_plus.h:147-150).  One streaming pass.
__global__ __launch_bounds__(256) void sp_fold_colterm_kernel(long long nnz, const int *__restrict__ indices,
                                                               const float *__restrict__ data, const float *__restrict__ Y,
                                                               float *__restrict__ out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const float y = Y[indices[i]];
        out[i] = (y != 0.f) ? data[i] / y : 0.f;
    }
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
    size_t ws_fold_bytes;   // scaled copy of m2_data when the column term is folded in (0 otherwise)
    size_t ws_order_bytes;  // bucket counters + work[n] + order[n] + row descriptors
    size_t ws_desc_offset;  // of the descriptors inside that block
    int nb_log2;            // sparse path bitmap bits (log2)
    size_t ws_total;        // header + gU + fold scratch + order scratch
    bool fold;
    bool ordered;
};

constexpr size_t WS_QUEUE_BYTES = 256;   // [0,8) row queue | [64,160) phase counters | [176,188) column-term minima
constexpr size_t WS_PHASE_OFFSET = 64;
constexpr size_t WS_YMIN_OFFSET = 176;
static_assert(WS_PHASE_OFFSET + PH_N * 8 <= WS_YMIN_OFFSET && WS_YMIN_OFFSET + 12 <= WS_QUEUE_BYTES, "workspace header layout");
constexpr size_t LDS_LIMIT = 160 * 1024;

size_t lds_fixed_bytes(int T, int NT) {
    // region A (table / bitmap) + region X (segment arrays | work items + 4 histograms) + hist + wsum + sh + ph
    const size_t xb = std::max<size_t>((size_t)16 * NT + 256, (size_t)ITEM_CAP * 16 + 4096);
    return (size_t)T * 8 + xb + 256 * 4 + 64 * 4 + 32 * 4 + 16 * 8;
}

int make_config(const sp_knn_args *a, int n_cus, Config *c) {
    int NT = a->threads_per_wg ? a->threads_per_wg : 1024;   // measured best on MI355X (16 waves/CU hide the LDS/HBM round trips)
    if (NT != 256 && NT != 512 && NT != 768 && NT != 1024) return fail(SP_EINVAL, "threads_per_wg must be 256, 512, 768 or 1024 (got %d)", NT);
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
    cap &= ~1LL;   // the kernel clears the buffer with 16-byte stores
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
    // product-form epilogue  val = xy / (l * X[t] * Y[c])  (cosine, asymmetric cosine, rp3beta without shrink):
    // Y is divided into the m2 values once per call, the kernel then needs no column-term gathers at all
    c->fold = !(a->flags & SP_FLAG_NO_FOLD) && a->l1 == 0.f && a->a1 == 1.f && a->stabilized_shrink == 0.f &&
              a->bayesian_shrink == 0.f && ((a->l2 != 0.f) != (a->l3 != 0.f)) && a->nnz_m2 > 0;
    c->ws_fold_bytes = c->fold ? (((size_t)a->nnz_m2 * 4 + 255) & ~(size_t)255) : 0;
    c->ordered = !(a->flags & (SP_FLAG_STATIC_SCHED | SP_FLAG_NO_ROW_ORDER)) && a->n_targets > num_wgs;
    // 512 B of bucket counters | work[n] | order[n] | (32-byte aligned) desc[n] of 32 B
    c->ws_desc_offset = (512 + (size_t)a->n_targets * 8 + 31) & ~(size_t)31;
    c->ws_order_bytes = (c->ws_desc_offset + (size_t)a->n_targets * 32 + 255) & ~(size_t)255;
    // sparse path: one bit per column while the columns fit region A, else columns alias modulo the bitmap size
    int nb = 10;
    while (nb < logT + 6 && (1LL << nb) < (long long)a->n_output_cols) ++nb;
    c->nb_log2 = nb;
    c->ws_total = WS_QUEUE_BYTES + ((c->ws_gu_bytes + 255) & ~(size_t)255) + c->ws_fold_bytes + c->ws_order_bytes;
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
    if (a->nnz_m2 >= (1LL << 30))
        return fail(SP_EINVAL, "nnz(m2) = %lld: this build addresses m2 with 32-bit byte offsets and needs nnz(m2) < 2^30",
                    (long long)a->nnz_m2);
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

    // minima of the column-term vectors feed the gather-free upper bound (Epi::upper); it is sound only
    // when every weight / shrink is non-negative (NaN parameters fail the comparisons and disable it)
    const bool bound_ok = (a->l1 >= 0.f) && (a->l2 >= 0.f) && (a->l3 >= 0.f) && (a->t1 >= 0.f) && (a->t2 >= 0.f) &&
                          (a->stabilized_shrink >= 0.f) && (a->bayesian_shrink >= 0.f);
    float *ymin_dev = (float *)(ws + WS_YMIN_OFFSET);
    float *folded = nullptr;
    if (c.fold) {
        folded = (float *)(ws + WS_QUEUE_BYTES + ((c.ws_gu_bytes + 255) & ~(size_t)255));
        hipLaunchKernelGGL(sp_fold_colterm_kernel, dim3(256 * 8), dim3(256), 0, stream, (long long)a->nnz_m2, a->m2_indices,
                           a->m2_data, a->l2 != 0.f ? a->Ycosine : a->Ydepop, folded);
        HIP_TRY(hipGetLastError());
    } else if (bound_ok && (a->l1 != 0.f || a->l2 != 0.f || a->l3 != 0.f)) {
        hipLaunchKernelGGL(sp_colterm_min_kernel, dim3(1), dim3(1024), 0, stream, a->n_output_cols,
                           a->l1 != 0.f ? a->Ytversky : nullptr, a->l2 != 0.f ? a->Ycosine : nullptr,
                           a->l3 != 0.f ? a->Ydepop : nullptr, ymin_dev);
        HIP_TRY(hipGetLastError());
    }

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
    {
        unsigned char *ob = ws + WS_QUEUE_BYTES + ((c.ws_gu_bytes + 255) & ~(size_t)255) + c.ws_fold_bytes;
        unsigned *bucket_count = (unsigned *)ob;            // [32]
        unsigned *bucket_base = bucket_count + 32;          // [32] + [1] flag
        unsigned *work = (unsigned *)(ob + 512);            // [n]
        int *order = (int *)(work + a->n_targets);          // [n]
        int4 *desc = (int4 *)(ob + c.ws_desc_offset);       // [2n]
        HIP_TRY(hipMemsetAsync(ob, 0, 512, stream));
        const int waves_per_block = 256 / 64;
        hipLaunchKernelGGL(sp_row_work_kernel, dim3((a->n_targets + waves_per_block - 1) / waves_per_block), dim3(256), 0, stream,
                           a->n_targets, a->targets, a->m1_indices, a->m1_indptr, a->m2_indptr, work, bucket_count);
        if (c.ordered) {
            hipLaunchKernelGGL(sp_bucket_base_kernel, dim3(1), dim3(64), 0, stream, bucket_count, bucket_base);
            hipLaunchKernelGGL(sp_row_order_kernel, dim3((a->n_targets + 255) / 256), dim3(256), 0, stream, a->n_targets, work, bucket_base, order);
        }
        hipLaunchKernelGGL(sp_row_desc_kernel, dim3((a->n_targets + 255) / 256), dim3(256), 0, stream, a->n_targets, a->targets,
                           a->m1_indptr, work, c.ordered ? bucket_base + 32 : nullptr, order, desc);
        HIP_TRY(hipGetLastError());
        kp.desc = desc;
    }
    kp.m2_bytes = (unsigned)((size_t)a->nnz_m2 * 4);
    kp.nb_log2 = c.nb_log2;
    kp.hash_fill = c.hash_fill;
    kp.static_sched = (a->flags & SP_FLAG_STATIC_SCHED) ? 1 : 0;
    kp.ymin = ymin_dev;
    kp.bound_ok = bound_ok ? 1 : 0;
    kp.sparse_path = (a->flags & SP_FLAG_NO_SPARSE_PATH) ? 0 : 1;
    kp.fold = c.fold ? 1 : 0;
    if (c.fold) kp.m2_data = folded;
    kp.phase_cycles = timed ? (unsigned long long *)(ws + WS_PHASE_OFFSET) : nullptr;   // inside the zeroed queue block
    kp.dbg = (int)a->reserved[0];

    if (c.NT == 256) rc = c.u_lds ? launch_rows<256, true>(kp, c, stream) : launch_rows<256, false>(kp, c, stream);
    else if (c.NT == 512) rc = c.u_lds ? launch_rows<512, true>(kp, c, stream) : launch_rows<512, false>(kp, c, stream);
    else if (c.NT == 768) rc = c.u_lds ? launch_rows<768, true>(kp, c, stream) : launch_rows<768, false>(kp, c, stream);
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
        const unsigned long long *phc = (const unsigned long long *)(qb + WS_PHASE_OFFSET);
        static_assert(PH_N == 12, "sp_knn_args::phase_cycles has 12 entries");
        for (int i = 0; i < PH_N; ++i) a->phase_cycles[i] = (int64_t)phc[i];
        a->passes_total = (int32_t)phc[CT_PASSES];
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
    a->num_wgs_used = d.num_wgs_used;
    memcpy(a->phase_cycles, d.phase_cycles, sizeof(a->phase_cycles));
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
