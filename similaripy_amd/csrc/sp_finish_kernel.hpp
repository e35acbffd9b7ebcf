// sp_finish_kernel.hpp — the dense end of a sparse-kernel row, one WAVE per row.
//
// Profiling (profiles/r02_exp_*): the two sweeps of sp_knn_sparse_kernel keep all 16 waves of a workgroup busy, but the
// phases behind them — member accumulation, collision-set scan, final selection, write-out — are chains of LDS round
// trips and workgroup barriers in which most threads have nothing to do: 22 k of a C2 row's 74 k cycles, during which the
// CU does little else (its 160 KiB of LDS hold one workgroup).  With `KParams::defer` the row kernel stops after its last
// sweep: it appends the row's member pool (products of marked columns) and candidate buffer (single products above the
// cutoff) to a log in HBM, and this kernel finishes the rows of the batch afterwards, one wave per row and four
// independent rows per CU — no barriers, the latencies of different rows overlap:
//   accumulate   the members go through a wave-private hash table in LDS ({column + 1 : sum}, 64-bit compare-and-swap claims
//                a slot, the LDS float add accumulates: s_plus.h:112-117)
//   scan         sums above the row's cutoff (and not excluded by a MATRIX filter, s_plus.h:159-171) join the candidates
//   select       exact top-k of the candidates: MSD radix select (4 x 8 bits) with a wave-private histogram (s_plus.h:39-64)
//   write-out    epilogue on the winners (val = xy / den or the raw dot: the monotone variant), exact threshold test,
//                compaction into the row's slot, zero tail, count (s_plus.h:444-450)
// Rows whose members do not fit the table (TS slots) are left to the next launch of this kernel with a larger table.
#pragma once
#include "sp_common.hpp"

namespace {

template <int TS, int NWF>
__global__ __launch_bounds__(64 * NWF) void sp_knn_finish_kernel(const KParams p, FinRec *__restrict__ recs, const u64 *__restrict__ dlog, int q_begin,
                                                             int q_end_max, unsigned *__restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsmem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    u64 *tab = (u64 *)fsmem + (size_t)wave * TS;                                // [TS] hash table, then the candidates
    int *hist = (int *)(fsmem + (size_t)NWF * TS * 8) + wave * 256;             // [256]
    for (int i = lane; i < TS; i += 64) tab[i] = 0ull;
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    const int n_recs = min(q_end_max, (int)p.qcount[0]) - q_begin;
    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    const int k = p.k;
    const size_t slice = (size_t)p.dslice;
    for (;;) {
        int ri = 0;
        if (lane == 0) ri = (int)atomicAdd(counter, 1u);
        ri = __builtin_amdgcn_readfirstlane(ri);
        if (ri >= n_recs) break;
        FinRec *rec = recs + ri;
        const int state = __builtin_amdgcn_readfirstlane(rec->state);
        if (state != 1) continue;
        const int n_mem = __builtin_amdgcn_readfirstlane(rec->n_mem);
        if (n_mem > (TS / 4) * 3) continue;                                    // a launch with a larger table takes it
        int n_cand = __builtin_amdgcn_readfirstlane(rec->n_cand);
        const float cutx = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)rec->cut_bits));
        const int slot_i = __builtin_amdgcn_readfirstlane(rec->d0.x), t = __builtin_amdgcn_readfirstlane(rec->d0.y);
        const float den = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(rec->d1.y));
        const u64 *mem = dlog + (size_t)ri * slice;
        const u64 *cand = mem + p.dmem_cap;
        const int cand_cap = (int)(slice - (size_t)p.dmem_cap);

        // ---- accumulate: {column + 1 : sum} find-or-insert, linear probing ----
        for (int j0 = 0; j0 < n_mem; j0 += 64) {
            const int j = j0 + lane;
            const u64 e = (j < n_mem) ? mem[j] : 0ull;
            if (e != 0ull) {
                const unsigned key = (unsigned)(e >> 32);
                unsigned h = hash_bits((int)key, 2654435761u, 0) >> (32 - __builtin_ctz(TS));
                for (int probe = 0; probe < TS; ++probe) {
                    const u64 old = atomicCAS(&tab[h], 0ull, e);
                    if (old == 0ull) break;                                                   // claimed, the product is in it
                    if ((unsigned)(old >> 32) == key) { atomicAdd((float *)&tab[h], __uint_as_float((unsigned)e)); break; }
                    h = (h + 1u) & (unsigned)(TS - 1);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (lanes talk through LDS without barriers: no store-to-load forwarding across phases)
        // ---- scan: complete sums above the cutoff join the candidates (in LDS behind... they are collected first in registers'
        // order: the table is read and zeroed slot by slot, survivors are written to the front part of the table that has
        // already been scanned) ----
        // Survivors of the scan are at most as many as slots scanned so far, so writing them to tab[0..) never overtakes
        // the read position.
        int f0 = 0, f1 = 0;
        if (p.filter_mode == SP_SEL_MATRIX) { f0 = __builtin_amdgcn_readfirstlane(p.f_indptr[t]); f1 = __builtin_amdgcn_readfirstlane(p.f_indptr[t + 1]); }
        int n_sur = 0;                                                               // (wave-uniform)
        for (int s0 = 0; s0 < TS; s0 += 64) {
            const u64 e = tab[s0 + lane];
            tab[s0 + lane] = 0ull;
            bool want = false;
            unsigned col = 0;
            float x = 0.f;
            if (e != 0ull) {
                col = (unsigned)(e >> 32) - 1u;
                x = __uint_as_float((unsigned)e);
                want = !(x <= cutx);
                if (want && p.filter_mode == SP_SEL_MATRIX && range_has(p.f_indices, f0, f1, (int)col)) want = false;
            }
            const u64 m = __ballot(want);
            if (want) tab[n_sur + mbcnt64(m)] = ((u64)fkey(x) << 32) | (u64)col;
            n_sur += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- candidates: the sweep's single products behind the scan's survivors ----
        int n_all = n_sur;
        bool overflow = false;
        for (int j0 = 0; j0 < n_cand; j0 += 64) {
            const int j = j0 + lane;
            const u64 e = (j < n_cand) ? cand[j] : 0ull;
            const u64 m = __ballot(e != 0ull);
            if (n_all + __popcll(m) > TS) { overflow = true; break; }                // (uniform)
            if (e != 0ull) tab[n_all + mbcnt64(m)] = e;
            n_all += __popcll(m);
        }
        (void)cand_cap;
        if (overflow) {
            // more candidates than the table holds (cannot happen while the row kernel's buffer is no larger than TS): generic kernel
            for (int i = lane; i < TS; i += 64) tab[i] = 0ull;
            if (lane == 0) {
                const unsigned g = atomicAdd(&p.qcount[1], 1u);
                p.desc_g[2 * (size_t)g] = rec->d0;
                p.desc_g[2 * (size_t)g + 1] = rec->d1;
                rec->state = 0;
            }
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- exact top-k: MSD radix select on the order-preserving key (the high half of an entry) ----
        unsigned prefix = 0u;
        int need = k;
        const bool select = n_all > k;
        if (select) {
            for (int shift = 24; shift >= 0; shift -= 8) {
                const unsigned hmask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
                for (int j = lane; j < n_all; j += 64) {
                    const unsigned key = (unsigned)(tab[j] >> 32);
                    if (((key ^ prefix) & hmask) == 0u) atomicAdd(&hist[(key >> shift) & 255u], 1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // lane L owns bins 255-4L .. 252-4L (lanes ascend as digits descend)
                const int4 c4 = *(const int4 *)&hist[252 - 4 * lane];
                *(int4 *)&hist[252 - 4 * lane] = make_int4(0, 0, 0, 0);
                const int c0 = c4.w, c1 = c4.z, c2 = c4.y, c3 = c4.x;
                const int s = c0 + c1 + c2 + c3;
                const int incl = wave_incl_scan_dpp(s);
                const int excl = incl - s;
                const bool mine = excl < need && need <= incl;                       // exactly one lane
                int d = 0, r = 0;
                if (mine) {
                    const int b0 = 255 - 4 * lane;
                    r = need - excl;
                    if (r <= c0) d = b0;
                    else if (r <= c0 + c1) { d = b0 - 1; r -= c0; }
                    else if (r <= c0 + c1 + c2) { d = b0 - 2; r -= c0 + c1; }
                    else { d = b0 - 3; r -= c0 + c1 + c2; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int src = (int)__builtin_ctzll(__ballot(mine));
                d = __builtin_amdgcn_readlane(d, src);
                need = __builtin_amdgcn_readlane(r, src);
                prefix |= (unsigned)d << shift;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- write-out: epilogue on the winners, exact threshold test, compaction, zero tail ----
        const long long o = (long long)slot_i * (long long)k;
        int n_out = 0, eq_taken = 0;                                                 // (wave-uniform)
        for (int j0 = 0; j0 < n_all; j0 += 64) {
            const int j = j0 + lane;
            const u64 it = (j < n_all) ? tab[j] : 0ull;
            if (j < n_all) tab[j] = 0ull;
            const unsigned key = (unsigned)(it >> 32);
            bool win = it != 0ull;
            if (select && win) win = key >= prefix;
            const u64 meq = __ballot(select && win && key == prefix);
            if (select && win && key == prefix) win = (eq_taken + mbcnt64(meq)) < need;      // the k-th place: first come, first kept
            eq_taken += __popcll(meq);
            const float xv = funkey(key);
            float val = xv;
            if (any_norm) val = (den != 0.f) ? xv / den : 0.f;
            const bool keep = win && (val >= p.threshold);
            const u64 mk = __ballot(keep);
            if (keep) {
                const long long q = o + n_out + mbcnt64(mk);
                if (p.rows) p.rows[q] = t;
                p.cols[q] = (int)(unsigned)(it & 0xFFFFFFFFull);
                p.values[q] = val;
            }
            n_out += __popcll(mk);
        }
        for (int j = n_out + lane; j < k; j += 64) {
            if (p.rows) p.rows[o + j] = 0;
            p.cols[o + j] = 0;
            p.values[o + j] = 0.f;
        }
        if (lane == 0) {
            if (p.counts) p.counts[slot_i] = n_out;
            rec->state = 2;
            if (p.phase_cycles) atomicAdd(&p.phase_cycles[CT_ROWS_SPARSE], 1ull);
        }
    }
}

}  // namespace
