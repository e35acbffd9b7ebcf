// sp_sddmm_kernel.hpp — target_cols = <sparse matrix> as a SAMPLED product (SDDMM).
//
// The reference accumulates the whole row of m1 x m2 and then throws away every candidate column that is not listed in the target
// row (s_plus.h:175-188: a lower_bound per candidate).  When the lists are short next to the row's products — a handful of
// (user, item) pairs to score against 40 k products per row — only the listed entries are worth computing:
//     value(t, c) = epilogue( dot(m1 row t, m2 column c) )      for c in target_cols[t]
// m2's column c is row c of m2^T: for the `matrix2=None` call that is m1 itself (no transpose is built at all), for a CSC matrix1 it is
// the m1 the library builds anyway, for an explicit matrix2 its transpose is built once per call (sp_transpose.hpp).
//
// One WAVE per target row (persistent, atomic slot queue).  The row's entries go into an LDS hash (column of m1 -> value; duplicates
// add up, order inside the row does not matter), in chunks of SD_LCAP entries when the row is longer.  Then one LANE per listed column:
// it walks row c of m2^T, probes the hash, and adds up the products — 64 listed columns in flight per wave.  A column without a common
// entry is no candidate (the reference lists a column on its first product, s_plus.h:112-117); the others go through the epilogue of
// s_plus.h:129-156 and the threshold into a wave-private candidate list, trimmed to the k best (bit-wise search for the k-th key)
// whenever it fills.  Slot layout of the output as everywhere: first n entries real, tail (0, 0, 0.0).
#pragma once
#include "sp_common.hpp"

namespace {

constexpr int SD_LCAP = 512;           // entries of row t per hash build
constexpr int SD_HS = 2 * SD_LCAP;     // hash slots (power of two)
constexpr int SD_CAP = 512;            // candidate list entries per wave (k <= SD_CAP / 2)
constexpr int SD_WAVES = 4;            // waves per workgroup
constexpr int SD_KMAX = SD_CAP / 2;
__host__ __device__ constexpr size_t sd_lds_bytes() { return (size_t)SD_WAVES * ((size_t)SD_HS * 8 + (size_t)SD_CAP * 8); }

struct SddmmParams {
    int n_targets;
    const int *targets;
    const float *m1_data; const int *m1_indices; const int *m1_indptr;
    const float *mt_data; const int *mt_indices; const int *mt_indptr;      // m2^T
    const int *t_indptr; const int *t_indices;                              // target selector (rows by absolute m1 row id)
    int filter_mode; const int *f_indptr; const int *f_indices;             // MATRIX filter (optional)
    const unsigned char *col_keep;                                          // ARRAY selectors on a device-built m2 (optional)
    const float *Xtv, *Ytv, *Xcos, *Ycos, *Xdep, *Ydep;
    float a1, l1, l2, l3, t1, t2, stab, bayes, threshold;
    int k;
    int *rows; int *cols; float *values; int *counts;
    unsigned *queue;
};

// keep the k largest of buf[0, n) (n <= SD_CAP, 8 entries per lane in registers): bit-wise search for the k-th largest key
__device__ __forceinline__ int sd_wave_topk(u64 *buf, int n, int k, int lane) {
    if (n <= k) return n;
    u64 e[SD_CAP / 64];
    unsigned key[SD_CAP / 64];
    bool valid[SD_CAP / 64];
#pragma unroll
    for (int j = 0; j < SD_CAP / 64; ++j) {
        const int i = lane + 64 * j;
        valid[j] = i < n;
        e[j] = valid[j] ? buf[i] : 0ull;
        key[j] = (unsigned)(e[j] >> 32);
    }
    unsigned T = 0u;
    for (int b = 31; b >= 0; --b) {
        const unsigned cand = T | (1u << b);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < SD_CAP / 64; ++j) cnt += __popcll(__ballot(valid[j] && key[j] >= cand));
        if (cnt >= k) T = cand;      // uniform
    }
    int n_gt = 0;
#pragma unroll
    for (int j = 0; j < SD_CAP / 64; ++j) n_gt += __popcll(__ballot(valid[j] && key[j] > T));
    const int need_eq = k - n_gt;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int pos = 0;
#pragma unroll
    for (int j = 0; j < SD_CAP / 64; ++j) {
        const bool w = valid[j] && key[j] > T;
        const u64 m = __ballot(w);
        if (w) buf[pos + mbcnt64(m)] = e[j];
        pos += __popcll(m);
    }
    int eq_seen = 0;
#pragma unroll
    for (int j = 0; j < SD_CAP / 64; ++j) {
        const bool w = valid[j] && key[j] == T;
        const u64 m = __ballot(w);
        const int rank = eq_seen + mbcnt64(m);
        if (w && rank < need_eq) buf[pos + rank] = e[j];
        eq_seen += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return k;
}

__global__ __launch_bounds__(64 * SD_WAVES) void sp_sddmm_kernel(const SddmmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *hkey = (int *)(smem + (size_t)wave * ((size_t)SD_HS * 8 + (size_t)SD_CAP * 8));
    float *hval = (float *)(hkey + SD_HS);
    u64 *cand = (u64 *)(hval + SD_HS);
    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);

    for (;;) {
        int slot = 0;
        if (lane == 0) slot = (int)atomicAdd(p.queue, 1u);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= p.n_targets) break;
        const int t = p.targets[slot];
        const int r0 = p.m1_indptr[t], r1 = p.m1_indptr[t + 1];
        const int g0 = p.t_indptr[t], g1 = p.t_indptr[t + 1];
        int f0 = 0, f1 = 0;
        if (p.filter_mode == SP_SEL_MATRIX) { f0 = p.f_indptr[t]; f1 = p.f_indptr[t + 1]; }
        Epi epi;
        epi.a1 = p.a1; epi.l1 = p.l1; epi.l2 = p.l2; epi.l3 = p.l3; epi.t1 = p.t1; epi.t2 = p.t2;
        epi.stab = p.stab; epi.bayes = p.bayes; epi.threshold = p.threshold; epi.any = any_norm; epi.bound = false; epi.cut_ok = false; epi.bA = 0.f; epi.bB = 0.f;
        epi.xtv = (p.l1 != 0.f) ? p.Xtv[t] : 0.f;
        epi.xcos = (p.l2 != 0.f) ? p.Xcos[t] : 0.f;
        epi.xdep = (p.l3 != 0.f) ? p.Xdep[t] : 0.f;
        const int n_chunks = (r1 - r0 + SD_LCAP - 1) / SD_LCAP;      // 0: an empty row (no candidates at all)
        auto build = [&](int c0, int c1) {      // hash of row t's entries [c0, c1)
            for (int i = lane; i < SD_HS; i += 64) { hkey[i] = -1; hval[i] = 0.f; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = c0 + lane; i < c1; i += 64) {
                const int u = p.m1_indices[i];
                const float v = p.m1_data[i];
                unsigned h = hash_bits(u, 2654435761u, 22) & (unsigned)(SD_HS - 1);
                for (;;) {
                    const int old = atomicCAS(&hkey[h], -1, u);
                    if (old == -1 || old == u) { atomicAdd(&hval[h], v); break; }
                    h = (h + 1u) & (unsigned)(SD_HS - 1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        if (n_chunks == 1) build(r0, r1);
        int n_cand = 0;
        if (n_chunks > 0) {
            for (int gb = g0; gb < g1; gb += 64) {
                const int gi = gb + lane;
                int c = (gi < g1) ? p.t_indices[gi] : -1;
                if (c >= 0 && p.col_keep != nullptr && p.col_keep[c] == 0) c = -1;                                   // ARRAY selectors
                if (c >= 0 && p.filter_mode == SP_SEL_MATRIX && range_has(p.f_indices, f0, f1, c)) c = -1;           // MATRIX filter (s_plus.h:159-171)
                float acc = 0.f;
                bool touched = false;
                const int e0 = (c >= 0) ? p.mt_indptr[c] : 0, e1 = (c >= 0) ? p.mt_indptr[c + 1] : 0;
                for (int ch = 0; ch < n_chunks; ++ch) {
                    if (n_chunks > 1) build(r0 + ch * SD_LCAP, min(r1, r0 + (ch + 1) * SD_LCAP));
                    for (int e = e0; e < e1; ++e) {
                        const int u = p.mt_indices[e];
                        unsigned h = hash_bits(u, 2654435761u, 22) & (unsigned)(SD_HS - 1);
                        for (;;) {
                            const int kk = hkey[h];
                            if (kk == u) { acc = __builtin_fmaf(hval[h], p.mt_data[e], acc); touched = true; break; }
                            if (kk == -1) break;
                            h = (h + 1u) & (unsigned)(SD_HS - 1);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                bool keep = false;
                float val = 0.f;
                if (touched) {
                    const float ytv = (p.l1 != 0.f) ? p.Ytv[c] : 0.f, ycos = (p.l2 != 0.f) ? p.Ycos[c] : 0.f, ydep = (p.l3 != 0.f) ? p.Ydep[c] : 0.f;
                    val = epi(acc, ytv, ycos, ydep);
                    keep = val >= p.threshold;
                }
                // (room for this batch: trim to the k best first)
                if (n_cand + 64 > SD_CAP) n_cand = sd_wave_topk(cand, n_cand, p.k, lane);
                const u64 m = __ballot(keep);
                if (keep) cand[n_cand + mbcnt64(m)] = ((u64)fkey(val) << 32) | (u64)(unsigned)c;
                n_cand += __popcll(m);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            n_cand = sd_wave_topk(cand, n_cand, p.k, lane);
        }
        const long long o = (long long)slot * (long long)p.k;
        for (int j = lane; j < p.k; j += 64) {
            int r = 0, c = 0;
            float v = 0.f;
            if (j < n_cand) { const u64 it = cand[j]; r = t; c = (int)(unsigned)(it & 0xFFFFFFFFull); v = funkey((unsigned)(it >> 32)); }
            if (p.rows) p.rows[o + j] = r;
            p.cols[o + j] = c;
            p.values[o + j] = v;
        }
        if (lane == 0 && p.counts) p.counts[slot] = n_cand;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

}  // namespace
