// sp_wave_kernel.hpp — LIGHT rows, one WAVE per row: the user-scoring shape of BASELINE configs[4]
// (dot_product(urm, W.T, filter_cols=urm): 64 m1 entries x ~100-element m2 rows = 6.4 k products over 100 k columns, k = 100).
//
// The workgroup-per-row kernel (sp_sparse_kernel.hpp) spends a light row's time at barriers and on memory round trips nobody
// overlaps: 59 k cycles for 6.4 k products, three 4-wave workgroups per CU (round 3: 3.9 B/clk/CU against the 13-16 B/clk a CU
// can pull).  Here a row belongs to ONE wave64 and the workgroup IS that wave: no barrier anywhere, every counter a scalar
// register, 13.3 KB of LDS per wave, i.e. TWELVE independent rows in flight per CU (eleven until round 6), and each wave keeps several trips of its row in
// flight.
//
// Same algorithm as the monotone variant of the sparse kernel (s_plus.h:71-127 dense sums[] -> column BITMAP + two sweeps;
// s_plus.h:39-64 heap -> radix selection; s_plus.h:129-156 epilogue on the winners; s_plus.h:159-171 MATRIX filter):
//   sweep 1 (column ids)      one bit per output column (n_cols <= 2^17; beyond, columns alias modulo 2^17 and an aliased column is
//                             marked like a repeated one — summed where it need not be, still exact); a product that finds its bit set marks its
//                             column in the 8 k-bit collision bitmap (columns alias modulo its size: an aliased column is only
//                             summed where it need not be);
//   clear + rank prefix       the bitmap's storage becomes [collision set 1024 slots | member pool | U];
//   sweep 2 (ids + values)    x = value * m1 value; products of marked columns -> member pool; every other product is the only
//                             one of its column and goes to the candidate buffer U (256 entries) iff x beats the running k-th
//                             value; a full U is cut back to its k largest by a wave-local bit-wise search (wave_select);
//   accumulate, drain         members find their slot by the rank of their column's bit (32-bit compare-and-swap claims, float add),
//                             complete sums above the cutoff join U; excluded (filter) columns carry a -inf pseudo member;
//   select, write-out         exact top-k, epilogue val = xy / den (or the raw dot), threshold, compaction.
// Work items: one record per SEGMENT (m1 entry x its m2 row), written once per call by sp_row_items_wave_kernel, 64 per row.  The segments
// lie end to end on a virtual lane axis (a segment of `len` elements takes ceil(len / 4) lanes, NO gaps) and a trip is a window of 64 lanes
// of it — any number of pieces per trip: a user-scoring row of 100-element segments fills 64 lanes of 64, the two-piece trips of the
// workgroup-per-row kernel 50 (C5: 32 -> 25 trips per row, 26.1 -> 21.2 ms).  Lane i of the wave holds segment i's record; lane T holds trip T's
// 64 start marks and the number of marks before them, and every lane of a trip finds its segment by counting marks (two v_mbcnt) and
// fetches the segment's {B, D, m1 value} from its holder by ds_bpermute.  At most 63 trips.
// The kernel has its own queue: sp_row_desc_kernel sends it the sparse rows with at most 64 m1 entries and 10 k products, the other
// sparse rows run on the workgroup-per-row kernel in the same call; a row that fails here (collision set full) joins the generic queue.
#pragma once
#include "sp_common.hpp"

namespace {

// trips in flight per wave, sweep 1 / sweep 2.  Measured on the gap-free trips (C5 kernel ms, -DWV_D1= -DWV_D2= builds): 2/2 21.7, 3/3 20.68, 4/3 20.70,
// 4/4 20.75, 6/5 20.85, 8/6 (round 4's choice) 21.1, 10/6 21.1; user scoring over 10^6 items (m2 streamed from HBM) 14.06 - 14.13 at every depth
#ifndef WV_D1
#define WV_D1 4
#endif
#ifndef WV_D2
#define WV_D2 3
#endif
constexpr int WV_BM_LOG2 = 17;                                  // column bitmap: up to 2^17 bits
constexpr int WV_CBM_BYTES = 1024;                              // collision bitmap: 8192 bits
constexpr int WV_PRE_BYTES = 512;                               // u16 rank prefix per collision-bitmap word
constexpr int WV_CS_DIRECT = 384, WV_CS_OVER = 512;             // collision-set slots: [0, 384) by rank, 512 overflow slots behind them
constexpr int WV_CSN = WV_CS_DIRECT + WV_CS_OVER;
constexpr int WV_UCAP = 256;                                    // candidate buffer entries (four per lane)
constexpr int WV_KMAX = 128;                                    // k + 64 <= UCAP must hold with room to spare
constexpr int WV_OFF_A = WV_CBM_BYTES;
constexpr int WV_OFF_PRE = WV_OFF_A + WV_CSN * 8;               // the rank prefix lies INSIDE region A (it is built after sweep 1, when the bitmap is gone)
constexpr int WV_OFF_MP = WV_OFF_PRE + WV_PRE_BYTES;
// Region A is the column bitmap during sweep 1 and afterwards [collision set 7 KB | rank prefix 512 B | member pool | U 2 KB].  Four sizes:
//   12 624 B (100 992 columns): member pool 362 entries, 13 648 B of LDS per wave = TWELVE rows in flight per CU (12 x 13 648 = 163 776 of 163 840);
//   12 800 B (102 400 columns): member pool 384 entries, 13 824 B per wave = eleven rows per CU;
//   15 360 B (122 880 columns): member pool 704 entries, 16 384 B per wave = ten rows per CU;
//   16 384 B (131 072 columns): member pool 832 entries, 17 408 B per wave = nine rows per CU.
// (Round 6, late: rows in flight are what this kernel's time follows — padding its LDS to ten / eight rows per CU cost 8.7 % / 20 % — and
// 14 336 B per wave were eleven.  The twelfth comes from the rank prefix moving into region A and 128 rank-addressed slots less: a row
// of this kernel marks ~200 columns, the classification admits an expectation of 307.)
// The member pool is small on purpose: when a trip's members do not fit, the pool is folded into the collision set right away
// (wave_accumulate) and starts over — LDS per wave is what bounds the rows in flight, and those are what hides this kernel's latencies.
constexpr int WV_A_TIGHT = 12624, WV_A_SMALL = 12800, WV_A_MID = 15360, WV_A_LARGE = 16384;
__host__ __device__ constexpr int wv_mpcap(int a_bytes) { return (a_bytes - WV_CSN * 8 - WV_PRE_BYTES - WV_UCAP * 8) / 8; }
__host__ __device__ constexpr int wv_off_u(int a_bytes) { return WV_OFF_A + a_bytes - WV_UCAP * 8; }
__host__ __device__ constexpr int wv_lds_bytes(int a_bytes) { return WV_OFF_A + a_bytes; }
static_assert(wv_mpcap(WV_A_TIGHT) >= 256 + 64, "a trip's members (<= 256) fit an empty pool, the filter's pseudo members a fresh one");
static_assert(12 * wv_lds_bytes(WV_A_TIGHT) <= 160 * 1024 && 11 * wv_lds_bytes(WV_A_SMALL) <= 160 * 1024 && 10 * wv_lds_bytes(WV_A_MID) <= 160 * 1024 && 9 * wv_lds_bytes(WV_A_LARGE) <= 160 * 1024, "rows per CU");
static_assert((WV_CS_OVER & (WV_CS_OVER - 1)) == 0 && WV_OFF_PRE % 16 == 0 && WV_A_TIGHT % 16 == 0, "layout");
// the host's choice of the region (make_config, the launch)
__host__ __device__ constexpr int wv_region_bytes(int n_cols) { return n_cols <= 8 * WV_A_TIGHT ? WV_A_TIGHT : n_cols <= 8 * WV_A_SMALL ? WV_A_SMALL : n_cols <= 8 * WV_A_MID ? WV_A_MID : WV_A_LARGE; }

// Sweep 2 core for a collision bitmap of WV_CBM_BYTES at LDS offset 0 (sp_common.hpp's s2_core with this kernel's mask).
// Element j of a lane is real iff j < d: both masks come out cut to the real elements (the cuts in the asm: left to the compiler they
// end up on the VALU, a move, an and and a v_readfirstlane per mask half).
__device__ __forceinline__ void s2_core_w(const unsigned (&c)[4], const float (&v)[4], float segv, float cut, int d, float (&x)[4], u64 (&M)[4], u64 (&L)[4]) {
    unsigned a0, a1, a2, a3;
    u64 k0, k1, k2, k3;
    asm volatile(
        "v_lshrrev_b32 %[a0], 3, %[c0]\n\t"
        "v_lshrrev_b32 %[a1], 3, %[c1]\n\t"
        "v_lshrrev_b32 %[a2], 3, %[c2]\n\t"
        "v_lshrrev_b32 %[a3], 3, %[c3]\n\t"
        "v_and_b32 %[a0], 0x3fc, %[a0]\n\t"
        "v_and_b32 %[a1], 0x3fc, %[a1]\n\t"
        "v_and_b32 %[a2], 0x3fc, %[a2]\n\t"
        "v_and_b32 %[a3], 0x3fc, %[a3]\n\t"
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
        "v_cmp_lt_i32_e64 %[k0], 0, %[d]\n\t"
        "v_cmp_lt_i32_e64 %[k1], 1, %[d]\n\t"
        "v_cmp_lt_i32_e64 %[k2], 2, %[d]\n\t"
        "v_cmp_lt_i32_e64 %[k3], 3, %[d]\n\t"
        "s_and_b64 %[L0], %[L0], %[k0]\n\t"
        "s_and_b64 %[L1], %[L1], %[k1]\n\t"
        "s_and_b64 %[L2], %[L2], %[k2]\n\t"
        "s_and_b64 %[L3], %[L3], %[k3]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfe_u32 %[a0], %[a0], %[c0], 1\n\t"
        "v_bfe_u32 %[a1], %[a1], %[c1], 1\n\t"
        "v_bfe_u32 %[a2], %[a2], %[c2], 1\n\t"
        "v_bfe_u32 %[a3], %[a3], %[c3], 1\n\t"
        "v_cmp_ne_u32_e64 %[M0], 0, %[a0]\n\t"
        "v_cmp_ne_u32_e64 %[M1], 0, %[a1]\n\t"
        "v_cmp_ne_u32_e64 %[M2], 0, %[a2]\n\t"
        "v_cmp_ne_u32_e64 %[M3], 0, %[a3]\n\t"
        "s_and_b64 %[M0], %[M0], %[k0]\n\t"
        "s_and_b64 %[M1], %[M1], %[k1]\n\t"
        "s_and_b64 %[M2], %[M2], %[k2]\n\t"
        "s_and_b64 %[M3], %[M3], %[k3]\n\t"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),
          [x0] "=&v"(x[0]), [x1] "=&v"(x[1]), [x2] "=&v"(x[2]), [x3] "=&v"(x[3]),
          [M0] "=&s"(M[0]), [M1] "=&s"(M[1]), [M2] "=&s"(M[2]), [M3] "=&s"(M[3]),
          [L0] "=&s"(L[0]), [L1] "=&s"(L[1]), [L2] "=&s"(L[2]), [L3] "=&s"(L[3]),
          [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
        : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]),
          [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [sv] "v"(segv), [cut] "s"(cut), [d] "v"(d)
        : "memory", "scc");
}

// Wave-local selection: the n (> k) entries {key : column} of U[0, n) are cut back to those with the largest keys, compacted to the front,
// the tail zeroed.  Every lane holds four entries in registers; the k-th largest key is found by a BIT-WISE search — for every bit below the
// keys' common prefix, from the top: do k entries reach the threshold with this bit set? — in compares, scalar mask counts and adds only.
// (Round 4's MSD radix select went through an LDS histogram: four passes of atomics, fences and read-backs, 777 instructions and a dozen
// LDS round trips per call, four or five calls per row — a quarter of the instructions of a row that is bound by instruction issue;
// profiles/r05_exp_dropped.txt.)
//   exact  (the row's final top-k): exactly k entries are kept (ties at the k-th place resolved arbitrarily, as the reference's heap does);
//   !exact (a full U in mid-row): the search stops as soon as at most k + 32 entries reach the threshold and keeps all of them — a valid
//          running cutoff (k entries reach it) for a dozen iterations instead of up to 32.
// Returns the threshold key and the entries kept.  A real call (several call sites in the unrolled sweeps).
struct WaveSel { unsigned key; int kept; };
__device__ __attribute__((noinline)) WaveSel wave_select(u64 *U, int n, int k, int lane, bool exact) {
    u64 e[4];
    unsigned key[4];
    bool live[4];
    unsigned lmax = 0u, lmin_inv = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        live[j] = i < n;
        e[j] = live[j] ? U[i] : 0ull;
        key[j] = (unsigned)(e[j] >> 32);
        if (live[j]) { lmax = max(lmax, key[j]); lmin_inv = max(lmin_inv, ~key[j]); }
    }
    const unsigned hi = wave_max_u32(lmax), lo = ~wave_max_u32(lmin_inv);
    const unsigned diff = hi ^ lo;
    unsigned T = hi;
    int cntT = n;
    const int slack = exact ? 0 : 32;
    bool all_bits = true;           // the search ran to the last bit: T is the exact k-th largest key
    if (diff != 0u) {
        int b = 31 - __clz((int)diff);
        T = (b == 31) ? 0u : ((hi >> (b + 1)) << (b + 1));      // the keys' common prefix
        for (; b >= 0; --b) {
            const unsigned cand = T | (1u << b);
            const int cnt = (__popcll(__ballot(live[0] && key[0] >= cand)) + __popcll(__ballot(live[1] && key[1] >= cand))) +
                            (__popcll(__ballot(live[2] && key[2] >= cand)) + __popcll(__ballot(live[3] && key[3] >= cand)));
            if (cnt >= k) {      // uniform
                T = cand;
                cntT = cnt;
                if (cnt <= k + slack) { all_bits = (b == 0); break; }
            }
        }
    }
    // cntT entries reach T (>= k of them).  They are all kept when that is few enough (inexact, or exactly k); else T is the exact k-th
    // largest key (every bit searched, or all keys equal): the larger ones and as many of its equals as are needed
    const bool keep_all = cntT <= k + slack;
    int n_gt = 0;
    if (!keep_all) n_gt = (__popcll(__ballot(live[0] && key[0] > T)) + __popcll(__ballot(live[1] && key[1] > T))) +
                          (__popcll(__ballot(live[2] && key[2] > T)) + __popcll(__ballot(live[3] && key[3] > T)));
    int need_eq = keep_all ? 0x7FFFFFFF : k - n_gt;
    (void)all_bits;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int pos = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool gt = live[j] && key[j] > T;
        const u64 m = __ballot(gt);
        if (gt) U[pos + mbcnt64(m)] = e[j];
        pos += __popcll(m);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool eq = live[j] && key[j] == T;
        const u64 m = __ballot(eq);
        const int r = mbcnt64(m);
        if (eq && r < need_eq) U[pos + r] = e[j];
        const int taken = min(__popcll(m), need_eq);
        pos += taken;
        if (!keep_all) need_eq -= taken;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (i >= pos && i < n) U[i] = 0ull;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return WaveSel{T, pos};
}

// The rare side of a sweep-2 trip: its survivors do not fit U.  U is cut back to its k largest (their smallest is the new cutoff,
// applied to what is left of the trip), quad by quad: k + 64 <= WV_UCAP, so one quad always fits behind a selection.
// A real call (one site per unrolled trip body): values in, the new {entries in U, cutoff} out.
struct WaveUState { int ucnt; float cutx; };
// (The survivor masks are not passed: a lane's product that is no survivor arrives as -inf.  Mask arguments would be copies SGPR -> VGPR,
// and those make the compiler move the caller's whole mask arithmetic — every trip's, not only the rare one's — to the VALU.)
__device__ __attribute__((noinline)) WaveUState wave_push_slow(u64 *U, unsigned u_off, unsigned c0, unsigned c1, unsigned c2, unsigned c3, float x0, float x1, float x2,
                                                               float x3, int ucnt, float cutx, float cutx0, int k, int lane) {
    const unsigned c[4] = {c0, c1, c2, c3};
    const float x[4] = {x0, x1, x2, x3};
    u64 S[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) S[j] = __ballot(!(x[j] <= -__builtin_inff()));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ns = __popcll(S[j]);
        if (ns && ucnt + ns > WV_UCAP) {
            const WaveSel ws = wave_select(U, ucnt, k, lane, false);
            ucnt = ws.kept;
            cutx = fmaxf(cutx0, funkey(ws.key));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) if (jj >= j) S[jj] &= __ballot(!(x[jj] <= cutx));
            ns = __popcll(S[j]);
        }
        if (ns) { lds_push64(S[j], c[j], fkey(x[j]), ucnt, u_off); ucnt += ns; }
    }
    return WaveUState{ucnt, cutx};
}

// Members -> collision set: find-or-insert, {column + 1 : sum} slots, key word 0 = free.  The direct slot is the rank of the column's
// bit in the collision bitmap; a slot taken by another column (bits alias) sends the entry to a hashed start in the overflow half,
// then on linearly.  ONE wave owns the set: a 32-bit compare-and-swap on the key word claims the slot (or finds it the column's
// already), then the product goes in with the hardware float add — 0.33 lanes/clk on gfx950 (profiles/r02_lds_atomics_bench.txt),
// ~2 k cycles for a row's ~700 members, against five dependent round trips per pass of the 64-bit compare-and-swap form the 16-wave
// kernel needs (its slots are contended).  Returns true when the set is full (the row goes to the generic kernel).
__device__ __attribute__((noinline)) bool wave_accumulate(u64 *cs, const u64 *mpool, const unsigned char *cbm, const unsigned short *pre16, int mcnt, int lane) {
    auto next_slot = [&](unsigned h, unsigned key) __attribute__((always_inline)) -> unsigned {
        const unsigned dir = (unsigned)WV_CS_DIRECT, ovr = (unsigned)WV_CS_OVER;      // (512 overflow slots: a 9-bit hash)
        return (h < dir) ? dir + hash_bits((int)key, 2654435761u, 32 - 9) : dir + ((h - dir + 1u) & (ovr - 1u));
    };
    constexpr int JA = 4;
    unsigned *csw = (unsigned *)cs;                      // word 2h: the sum, word 2h + 1: the key
    bool failed = false;
    for (int base = 0; base < mcnt && !failed; base += JA * 64) {
        unsigned kk[JA], h[JA];
        float xx[JA];
        bool act[JA];
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int i = base + j * 64 + lane;
            const u64 e = (i < mcnt) ? mpool[i] : 0ull;
            kk[j] = (unsigned)(e >> 32);
            xx[j] = __uint_as_float((unsigned)e);
            act[j] = (e != 0ull);
            const unsigned cm = kk[j] - 1u;
            const unsigned wi = (cm >> 5) & (unsigned)(WV_CBM_BYTES / 4 - 1);
            const unsigned bw = ((const unsigned *)cbm)[wi];
            h[j] = (unsigned)pre16[wi] + (unsigned)__popc(bw & ((1u << (cm & 31u)) - 1u));
        }
        int rounds = 0;
        while (__ballot((act[0] | act[1]) | (act[2] | act[3]))) {
            unsigned r[JA];
#pragma unroll
            for (int j = 0; j < JA; ++j) r[j] = act[j] ? atomicCAS(&csw[2 * h[j] + 1], 0u, kk[j]) : 0u;
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                if (act[j]) {
                    if (r[j] == 0u || r[j] == kk[j]) { atomicAdd((float *)&csw[2 * h[j]], xx[j]); act[j] = false; }
                    else h[j] = next_slot(h[j], kk[j]);          // another column's slot
                }
            }
            if (++rounds > 4 * CS_MAXPROBE) { failed = true; break; }      // set full
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return failed;
}

template <int A_BYTES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void sp_knn_wave_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    unsigned char *cbm = smem;
    unsigned short *pre16 = (unsigned short *)(smem + WV_OFF_PRE);
    unsigned char *rA = smem + WV_OFF_A;
    u64 *cs = (u64 *)rA;
    u64 *mpool = (u64 *)(smem + WV_OFF_MP);
    constexpr int MPCAP = wv_mpcap(A_BYTES);
    constexpr unsigned U_OFF = (unsigned)wv_off_u(A_BYTES);
    constexpr int LDS_BYTES = wv_lds_bytes(A_BYTES);
    u64 *U = (u64 *)(smem + U_OFF);
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_indices, 0, (int)p.m2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_val = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_data, 0, (int)p.m2_bytes, 0x00020000);
    for (int i = lane; i < LDS_BYTES / 16; i += 64) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    const unsigned amask = (unsigned)((1 << (WV_BM_LOG2 - 3)) - 1) & ~3u;      // (n_cols <= 8 * A_BYTES: every column its own bit; more columns — the 16 KB region only — alias modulo 2^17)
    // phase timers (profiling passes only): lane 0 adds every interval straight to the global counters — no registers held for them
    const bool timing = (p.phase_cycles != nullptr) && lane == 0;
    u64 tmark = timing ? (u64)clock64() : 0;
#define WV_PHASE_END(which) do { if (timing) { const u64 _n = (u64)clock64(); atomicAdd(&p.phase_cycles[which], _n - tmark); tmark = _n; } } while (0)

    const int n_rows = (int)p.qcount[0];
    const int k = p.k;
    int q_next = 0;
    if (p.static_sched) q_next = (int)blockIdx.x;
    else { if (lane == 0) q_next = (int)atomicAdd(&p.queue[0], 1u); q_next = __builtin_amdgcn_readfirstlane(q_next); }

    for (;;) {
        const int q = q_next;
        if (q >= n_rows) break;
        // the next row's queue position is claimed now and arrives while this row runs
        int q_claim = 0;
        if (p.static_sched) q_claim = q + (int)gridDim.x;
        else if (lane == 0) q_claim = (int)atomicAdd(&p.queue[0], 1u);

        const int4 d0 = p.desc[2 * (size_t)q], d1 = p.desc[2 * (size_t)q + 1];
        const int slot = __builtin_amdgcn_readfirstlane(d0.x), t = __builtin_amdgcn_readfirstlane(d0.y);
        const int dw = __builtin_amdgcn_readfirstlane(d0.w);
        const int n1 = desc_n1(dw), n_tr = desc_n_trips(dw);
        const float den = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(d1.y));
        bool failed = (n_tr <= 0 || n_tr > 63 || slot >= p.items_rows);

        // segment records (sp_row_items_wave_kernel): lane i holds segment i = {byte offset of its m2 row, len, m1 value bits}, in the order the
        // segments are visited.  The segments lie end to end on the virtual lane axis — segment i starts at lane V_i = sum of ceil(len / 4)
        // before it — and lane l of trip T is place u = 64 T + l of it: with B = offset - 16 V and D = len + 4 V the lane's byte offset is
        // B + 16 u and it has D - 4 u elements left.  It belongs to segment #{i >= 1 : V_i - 1 < u}: the number of START MARKS (bit V_i - 1 of
        // an axis-long bitmap) below u.  The bitmap is built in LDS — one ds_or per segment, region A is clean between rows —, lane T takes
        // trip T's 64 marks and the number of marks before them, and the 512 bytes go back to zero.
        int4 seg = make_int4(0, 0, 0, 0);      // {B, D, m1 value bits, V}
        int f0 = 0, fl = 0;
        if (!failed) {
            const WaveSegRec r = ((const WaveSegRec *)(p.items_g + (size_t)slot * (size_t)p.items_stride))[lane];
            if (p.filter_mode == SP_SEL_MATRIX) { f0 = p.f_indptr[t]; fl = p.f_indptr[t + 1] - f0; }
            const int L = (r.len + 3) >> 2;
            const int V = wave_incl_scan_dpp(L) - L;
            if (r.len > 0) seg = make_int4(r.off4 - 16 * V, r.len + 4 * V, (int)r.vbits, V);
        }
        unsigned t_mlo, t_mhi;
        int t_cum;
        {
            if (lane >= 1 && seg.y != 0) { const unsigned g = (unsigned)seg.w - 1u; atomicOr((unsigned *)rA + (g >> 5), 1u << (g & 31u)); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint2 m = ((const uint2 *)rA)[lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            ((uint2 *)rA)[lane] = make_uint2(0u, 0u);
            t_mlo = m.x; t_mhi = m.y;
            const int cnt = __popc(m.x) + __popc(m.y);
            t_cum = wave_incl_scan_dpp(cnt) - cnt;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const int lane16 = lane * 16, nlane4 = -4 * lane;
        // x <= cutx0  =>  val(x) < threshold for sure (the exact test is repeated on the winners at write-out)
        float cutx0;
        if (!any_norm) cutx0 = __uint_as_float(RowCtx::funkey_inv_below(p.threshold));
        else {
            const float c0 = p.threshold * den;
            cutx0 = c0 - fabsf(c0) * 2e-6f - 1e-37f;
            if (!(c0 == c0)) cutx0 = -__builtin_inff();
        }
        float cutx = cutx0;
        int ucnt = 0, mcnt = 0;
        WV_PHASE_END(PH_SETUP);

        // one trip of the wave: this lane's byte offset into m2, the number of its real elements (<= 0: none), its m1 value
        auto trip_lane = [&](int T, int &vo, int &d, float &sv) __attribute__((always_inline)) {
            const int tl = min(T, 63);                 // (lane 63 never holds a trip: n_tr <= 63; trips behind the row's last find d <= 0 everywhere)
            const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)t_mlo, tl), mhi = (unsigned)__builtin_amdgcn_readlane((int)t_mhi, tl);
            const int cb = __builtin_amdgcn_readlane(t_cum, tl);
            const int j4 = ((int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) + cb) << 2;
            const int Bj = __builtin_amdgcn_ds_bpermute(j4, seg.x), Dj = __builtin_amdgcn_ds_bpermute(j4, seg.y);
            sv = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(j4, seg.z));
            vo = Bj + lane16 + T * 1024;
            d = Dj + nlane4 - T * 256;
            if (d <= 0) vo = (int)OOB_SOFFSET;         // lanes beyond the segments fetch nothing
#if SP_ABLATION
            if ((p.dbg & 512) && d > 0) vo = lane * 16 + (T & 3) * 1024;      // ablation: every trip reads the same 4 KB (cache hits only)
#endif
        };

        if (!failed) {
            // ---- MATRIX filter (s_plus.h:159-171): the row's excluded columns, requested before the first sweep ----
            int my_fc = -1;
            if (p.filter_mode == SP_SEL_MATRIX && lane < fl) my_fc = p.f_indices[f0 + lane];

            // ---- sweep 1: column ids only, four trips in flight ----
            {
                auto ld = [&](int T, u32x4 &ids, int &d) __attribute__((always_inline)) {
                    int vo; float sv;
                    trip_lane(T, vo, d, sv);
                    ids = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, vo, 0, 0);
                };
                auto body = [&](int T, const u32x4 &ids, int d) __attribute__((always_inline)) {
                    if (T >= n_tr) return;
#if SP_ABLATION
                    if (p.dbg & 8) { asm volatile("" ::"v"(ids)); return; }      // ablation: loads only
#endif
                    const unsigned c[4] = {ids.x, ids.y, ids.z, ids.w};
                    // padding at QUAD granularity (as s1_core8q): a lane without a real element ORs nothing and reports nothing; the lane
                    // that holds a piece's last elements treats its whole quad as real — the up to three ids behind the piece's end
                    // (the next m2 row's first ids, or 0 behind the array's end) set bits nobody asked for, which is safe: a column
                    // marked without a second product only takes the collision-set route, where the sum of its one product is exact
                    const unsigned o = d > 0 ? 1u : 0u;
                    const unsigned one[4] = {o, o, o, o};
                    unsigned seen[4];
                    s1_core<WV_OFF_A, true>(c, one, amask, seen);
                    // a few of a trip's products find their column's bit set; a LANE rarely has two: its column is then the sum of
                    // seen[j] * c[j] (v_mad_u32_u24: columns are below 2^17) and goes out in ONE masked atomic
                    const unsigned cnt = (seen[0] + seen[1]) + (seen[2] + seen[3]);
                    if (__ballot(cnt != 0u)) {
                        unsigned cs1;
                        asm("v_mul_u32_u24 %0, %1, %5\n\t"
                            "v_mad_u32_u24 %0, %2, %6, %0\n\t"
                            "v_mad_u32_u24 %0, %3, %7, %0\n\t"
                            "v_mad_u32_u24 %0, %4, %8, %0"
                            : "=&v"(cs1)
                            : "v"(seen[0]), "v"(seen[1]), "v"(seen[2]), "v"(seen[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
                        if (cnt == 1u) atomicOr((unsigned *)(cbm + ((cs1 >> 3) & (unsigned)(WV_CBM_BYTES - 4))), 1u << (cs1 & 31u));
                        if (__ballot(cnt > 1u)) {
                            if (cnt > 1u) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (seen[j]) atomicOr((unsigned *)(cbm + ((c[j] >> 3) & (unsigned)(WV_CBM_BYTES - 4))), 1u << (c[j] & 31u));
                            }
                        }
                    }
                };
                // D1 trips in flight.  The order "body of trip T, then the load of trip T + D1 into the registers it freed" is pinned
                // with scheduling barriers: left alone the compiler sinks the loads to the loop's end and waits for ALL of them
                // (s_waitcnt vmcnt(0)) at its top — a batch, not a pipeline
                constexpr int D1 = WV_D1;
                u32x4 b[D1];
                int e[D1];
#pragma unroll
                for (int i = 0; i < D1; ++i) ld(i, b[i], e[i]);
                __builtin_amdgcn_sched_barrier(0);
                for (int T = 0; T < n_tr; T += D1) {
#pragma unroll
                    for (int i = 0; i < D1; ++i) {
                        body(T + i, b[i], e[i]);
                        __builtin_amdgcn_sched_barrier(0);
                        ld(T + i + D1, b[i], e[i]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // excluded columns: marked in the collision bitmap, so all their products gather in the collision set
            if (p.filter_mode == SP_SEL_MATRIX) {
                if (my_fc >= 0) atomicOr((unsigned *)(cbm + (((unsigned)my_fc >> 3) & (unsigned)(WV_CBM_BYTES - 4))), 1u << ((unsigned)my_fc & 31u));
                for (int i = 64 + lane; i < fl; i += 64) {
                    const unsigned c = (unsigned)p.f_indices[f0 + i];
                    atomicOr((unsigned *)(cbm + ((c >> 3) & (unsigned)(WV_CBM_BYTES - 4))), 1u << (c & 31u));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            WV_PHASE_END(PH_SWEEP1);

            // ---- the bitmap has done its job: back to zero; rank prefix of the collision bitmap ----
            for (int i = lane; i < A_BYTES / 16; i += 64) ((int4 *)rA)[i] = make_int4(0, 0, 0, 0);
            {
                const int4 w4 = ((const int4 *)cbm)[lane];           // 256 words: four per lane
                const int p0 = __popc((unsigned)w4.x), p1 = p0 + __popc((unsigned)w4.y), p2 = p1 + __popc((unsigned)w4.z);
                const int tot = p2 + __popc((unsigned)w4.w);
                const int incl = wave_incl_scan_dpp(tot);
                const int ex = incl - tot;
                ((u64 *)pre16)[lane] = (u64)(unsigned)(ex & 0xFFFF) | ((u64)(unsigned)((ex + p0) & 0xFFFF) << 16) |
                                       ((u64)(unsigned)((ex + p1) & 0xFFFF) << 32) | ((u64)(unsigned)((ex + p2) & 0xFFFF) << 48);
                if (__builtin_amdgcn_readlane(incl, 63) > WV_CS_DIRECT) failed = true;      // more marked columns than direct slots
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // every excluded column gets a pseudo member of value -inf: its sum is then below any cutoff
            if (p.filter_mode == SP_SEL_MATRIX && !failed) {
                for (int i0 = 0; i0 < fl && !failed; i0 += 64) {
                    const int c = (i0 == 0) ? my_fc : ((i0 + lane < fl) ? p.f_indices[f0 + i0 + lane] : -1);
                    const u64 m = __ballot(c >= 0);
                    if (mcnt + __popcll(m) > MPCAP) { failed = __builtin_amdgcn_readfirstlane((int)wave_accumulate(cs, mpool, cbm, pre16, mcnt, lane)) != 0; mcnt = 0; }
                    if (c >= 0) mpool[mcnt + mbcnt64(m)] = ((u64)((unsigned)c + 1u) << 32) | (u64)0xFF800000u;
                    mcnt += __popcll(m);
                }
            }
            WV_PHASE_END(PH_SEGMENTS);
        }

        if (!failed) {
            // ---- sweep 2: ids + values, four trips in flight ----
            auto ld = [&](int T, u32x4 &ids, u32x4 &vals, int &d, float &sv) __attribute__((always_inline)) {
                int vo;
                trip_lane(T, vo, d, sv);
                int vo_i = vo;
#if SP_ABLATION
                if ((p.dbg & 32768) && d > 0) vo_i = lane * 16 + (T & 3) * 1024;      // ablation: sweep 2's id loads hit the cache (what would ids kept in registers gain?)
#endif
                ids = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, vo_i, 0, 0);
                vals = __builtin_amdgcn_raw_buffer_load_b128(rs_val, vo, 0, 0);
            };
            auto body = [&](int T, const u32x4 &ids, const u32x4 &vals, int d, float sv) __attribute__((always_inline)) {
                if (T >= n_tr || failed) return;
#if SP_ABLATION
                if (p.dbg & 16) { asm volatile("" ::"v"(ids), "v"(vals)); return; }      // ablation: loads only
#endif
                const unsigned c[4] = {ids.x, ids.y, ids.z, ids.w};
                const float v[4] = {__uint_as_float(vals.x), __uint_as_float(vals.y), __uint_as_float(vals.z), __uint_as_float(vals.w)};
                float x[4];
                u64 M[4], S[4];
                s2_core_w(c, v, sv, cutx, d, x, M, S);
                // products of marked columns: the member pool.  A trip nearly always holds some, a LANE rarely more than one: the first
                // member of every lane goes out in ONE push, second and later members of a lane in the rare pushes behind it
                const u64 Many = (M[0] | M[1]) | (M[2] | M[3]);
                if (Many) {
                    const u64 R1 = M[1] & M[0], R2 = M[2] & (M[0] | M[1]), R3 = M[3] & ((M[0] | M[1]) | M[2]);
                    const int n0 = __popcll(Many), n1 = __popcll(R1), n2 = __popcll(R2), n3 = __popcll(R3);
                    if (mcnt + (n0 + n1) + (n2 + n3) > MPCAP) {
                        // the pool is folded into the collision set right away and starts over
                        failed = __builtin_amdgcn_readfirstlane((int)wave_accumulate(cs, mpool, cbm, pre16, mcnt, lane)) != 0;
                        mcnt = 0;
                        if (failed) return;
                    }
                    const unsigned xm = mask_select(M[0], __float_as_uint(x[0]), mask_select(M[1], __float_as_uint(x[1]), mask_select(M[2], __float_as_uint(x[2]), __float_as_uint(x[3]))));
                    const unsigned cm = mask_select(M[0], c[0], mask_select(M[1], c[1], mask_select(M[2], c[2], c[3])));
                    lds_push64(Many, xm, cm + 1u, mcnt, (unsigned)WV_OFF_MP);
                    mcnt += n0;
                    if ((R1 | R2) | R3) {
                        if (n1) lds_push64(R1, __float_as_uint(x[1]), c[1] + 1u, mcnt, (unsigned)WV_OFF_MP);
                        mcnt += n1;
                        if (n2) lds_push64(R2, __float_as_uint(x[2]), c[2] + 1u, mcnt, (unsigned)WV_OFF_MP);
                        mcnt += n2;
                        if (n3) lds_push64(R3, __float_as_uint(x[3]), c[3] + 1u, mcnt, (unsigned)WV_OFF_MP);
                        mcnt += n3;
                    }
                }
                // every other product is the only one of its column: into U iff it beats the running k-th value (rare once the cutoff
                // has settled: the masks are only cut when there is something to cut)
                if ((S[0] | S[1]) | (S[2] | S[3])) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) S[j] &= ~M[j];
                    if ((S[0] | S[1]) | (S[2] | S[3])) {
                        const int n0 = __popcll(S[0]), n1 = __popcll(S[1]), n2 = __popcll(S[2]), n3 = __popcll(S[3]);
                        if (ucnt + (n0 + n1) + (n2 + n3) > WV_UCAP) {
                            const float ninf = -__builtin_inff();
                            const WaveUState r = wave_push_slow(U, U_OFF, c[0], c[1], c[2], c[3], __uint_as_float(mask_select(S[0], __float_as_uint(x[0]), __float_as_uint(ninf))),
                                                                __uint_as_float(mask_select(S[1], __float_as_uint(x[1]), __float_as_uint(ninf))),
                                                                __uint_as_float(mask_select(S[2], __float_as_uint(x[2]), __float_as_uint(ninf))),
                                                                __uint_as_float(mask_select(S[3], __float_as_uint(x[3]), __float_as_uint(ninf))), ucnt, cutx, cutx0, k, lane);
                            ucnt = __builtin_amdgcn_readfirstlane(r.ucnt);
                            cutx = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(r.cutx)));
                        } else {
                            if (n0) lds_push64(S[0], c[0], fkey(x[0]), ucnt, U_OFF);
                            ucnt += n0;
                            if (n1) lds_push64(S[1], c[1], fkey(x[1]), ucnt, U_OFF);
                            ucnt += n1;
                            if (n2) lds_push64(S[2], c[2], fkey(x[2]), ucnt, U_OFF);
                            ucnt += n2;
                            if (n3) lds_push64(S[3], c[3], fkey(x[3]), ucnt, U_OFF);
                            ucnt += n3;
                        }
                    }
                }
            };
            constexpr int D2 = WV_D2;
            u32x4 bi[D2], bv[D2];
            int e[D2];
            float sg[D2];
#pragma unroll
            for (int i = 0; i < D2; ++i) ld(i, bi[i], bv[i], e[i], sg[i]);
            __builtin_amdgcn_sched_barrier(0);
            for (int T = 0; T < n_tr; T += D2) {
#pragma unroll
                for (int i = 0; i < D2; ++i) {
                    body(T + i, bi[i], bv[i], e[i], sg[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    ld(T + i + D2, bi[i], bv[i], e[i], sg[i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            WV_PHASE_END(PH_SWEEP2);
        }

        if (!failed) {
            // ---- what is left in the member pool joins the collision set (the verdict is made uniform: counters and masks stay in SGPRs) ----
            failed = __builtin_amdgcn_readfirstlane((int)wave_accumulate(cs, mpool, cbm, pre16, mcnt, lane)) != 0;
            // the member pool is consumed: its storage (and the collision bitmap) go back to zero at the row's end
            WV_PHASE_END(PH_ACCUM);
        }

        if (!failed) {
            // ---- the collision set's complete sums above the cutoff join U (an excluded column's sum is -inf ... unless an infinite
            // product made it NaN: then the filter list decides) ----
            for (int base = 0; base < WV_CSN; base += 4 * 64) {
                u64 e[4];
                bool want[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = (base + j * 64 < WV_CSN) ? cs[base + j * 64 + lane] : 0ull;      // (the set's 896 slots: the last trip holds two quads)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (base + j * 64 < WV_CSN) cs[base + j * 64 + lane] = 0ull;
                    want[j] = (e[j] != 0ull) && !(__uint_as_float((unsigned)e[j]) <= cutx);
                }
                if (p.filter_mode == SP_SEL_MATRIX) {
                    bool odd = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) odd |= want[j] && (__uint_as_float((unsigned)e[j]) != __uint_as_float((unsigned)e[j]));
                    if (__ballot(odd)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (want[j] && (__uint_as_float((unsigned)e[j]) != __uint_as_float((unsigned)e[j])) &&
                                range_has(p.f_indices, f0, f0 + fl, (int)((unsigned)(e[j] >> 32) - 1u))) want[j] = false;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u64 m = __ballot(want[j]);
                    int ns = __popcll(m);
                    if (ns && ucnt + ns > WV_UCAP) {
                        const WaveSel ws = wave_select(U, ucnt, k, lane, false);
                        ucnt = __builtin_amdgcn_readfirstlane(ws.kept);      // (a call's results arrive in VGPRs: made uniform again, or every
                        cutx = fmaxf(cutx0, funkey((unsigned)__builtin_amdgcn_readfirstlane((int)ws.key)));      // count and mask of the sweeps moves to the VALU)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) if (jj >= j) want[jj] = want[jj] && !(__uint_as_float((unsigned)e[jj]) <= cutx);
                        m = __ballot(want[j]);
                        ns = __popcll(m);
                    }
                    if (ns) {
                        if (want[j]) U[ucnt + mbcnt64(m)] = ((u64)fkey(__uint_as_float((unsigned)e[j])) << 32) | (u64)((unsigned)(e[j] >> 32) - 1u);
                        ucnt += ns;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            WV_PHASE_END(PH_DRAIN);
            if (ucnt > k) ucnt = __builtin_amdgcn_readfirstlane(wave_select(U, ucnt, k, lane, true).kept);
            WV_PHASE_END(PH_SELECT);

            // ---- write-out: epilogue on the winners (s_plus.h:129-156 with the column term folded in: val = xy / den, or the raw
            // dot), exact threshold test, compaction to the front of the slot, zero padding behind (s_plus.h:444-450) ----
            const long long o = (long long)slot * (long long)k;
            int n_out = 0;
            for (int base = 0; base < ucnt; base += 64) {
                const int j = base + lane;
                const u64 it = (j < ucnt) ? U[j] : 0ull;
                const float xv = funkey((unsigned)(it >> 32));
                float val = xv;
                if (any_norm) val = (den != 0.f) ? xv / den : 0.f;
                const bool keep = (it != 0ull) && (val >= p.threshold);
                const u64 m = __ballot(keep);
                if (keep) {
                    const long long qo = o + n_out + mbcnt64(m);
                    if (p.rows) p.rows[qo] = t;
                    p.cols[qo] = (int)(unsigned)(it & 0xFFFFFFFFull);
                    p.values[qo] = val;
                }
                n_out += __popcll(m);
            }
            for (int j = n_out + lane; j < k; j += 64) {
                if (p.rows) p.rows[o + j] = 0;
                p.cols[o + j] = 0;
                p.values[o + j] = 0.f;
            }
            if (lane == 0 && p.counts) p.counts[slot] = n_out;
            if (timing) atomicAdd(&p.phase_cycles[CT_ROWS_SPARSE], 1ull);
        } else {
            // not a row for this kernel (no trip records, too many trips) or a pool overflowed: the generic kernel's queue takes it
            if (lane == 0) {
                const unsigned g = atomicAdd(p.qcount_g, 1u);
                p.desc_g[2 * (size_t)g] = make_int4(d0.x, d0.y, d0.z, n1);      // (without the record counts)
                p.desc_g[2 * (size_t)g + 1] = d1;
            }
            if (timing) atomicAdd(&p.phase_cycles[CT_ROWS_FALLBACK], 1ull);
        }
        // LDS back to clean: collision bitmap, what is left of the member pool / collision set, U
        ((int4 *)cbm)[lane] = make_int4(0, 0, 0, 0);
        for (int i = lane; i < A_BYTES / 16; i += 64) ((int4 *)rA)[i] = make_int4(0, 0, 0, 0);      // (U is its last 2 KB)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        q_next = __builtin_amdgcn_readfirstlane(q_claim);
        WV_PHASE_END(PH_OUTPUT);
    }
#undef WV_PHASE_END
}

}  // namespace
