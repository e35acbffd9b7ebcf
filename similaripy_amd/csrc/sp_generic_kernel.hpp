// sp_generic_kernel.hpp — rows that are NOT sparse (dense / colliding rows, tiny matrices, huge k) and rows the
// sparse kernel gave up on: LDS accumulator tile of T {column, partial dot} slots, direct-indexed when the column
// window is <= T wide, hashed (64-bit compare-and-swap claims a slot and deposits the first product) otherwise;
// rows whose candidates do not fit one tile are processed in column windows exactly like the reference's blocked
// path (s_plus.h:350-410), the top-k state carried across windows.
#pragma once
#include "sp_common.hpp"

namespace {

// BIG: nnz(m2) >= 2^30 — the m2 streams are addressed with 64-bit byte offsets (one more VGPR and a 64-bit add per load);
// the host launches this variant for such calls and sends every row to it (the sparse kernel's buffer loads stop at 4 GB).
template <int NT, bool U_LDS, bool BIG = false>
__global__ __launch_bounds__(NT) void sp_knn_generic_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T;

    // ---- LDS carve-up (single dynamic array) ----
    u64 *tab = (u64 *)smem;                     // [T]  {column id : partial dot product}
    int *seg_lo = (int *)(tab + T);             // [NT]   start of the (window's) slice of m2 row u
    int *seg_pre = seg_lo + NT;                 // [NT+64] exclusive prefix of slice lengths
    float *seg_v1 = (float *)(seg_pre + NT + 64);  // [NT] m1 value of the segment
    int *seg_hi = (int *)(seg_v1 + NT);         // [NT]   end of the slice (= start of the next window's)
    int *hist = seg_hi + NT;                    // [256]
    int *wsum = hist + 256;                     // [64]
    int *sh = wsum + 64;                        // [32]
    u64 *ph = (u64 *)(sh + 32);                 // [16] phase timers / event counters (lane 0 only)
    u64 *U = U_LDS ? (u64 *)(ph + 16) : (p.gU_g + (size_t)blockIdx.x * (size_t)p.cap);

    // Two formats of the tile.  Hashed windows: T slots of {column id, partial sum} (EMPTY64 when free).  Dense windows
    // (direct-indexed, every column owns its slot): 2*T 32-bit partial sums, EMPTY32 (a NaN pattern no sum of finite
    // products has) marking a column nobody touched — twice the columns per window, half the windows per row, and the
    // 32-bit LDS compare-and-swap is the faster one.  The tile is refilled only when consecutive windows differ in format.
    unsigned *tabw = (unsigned *)smem;          // [2*T] the same bytes as 32-bit sums
    constexpr unsigned EMPTY32 = 0xFFFFFFFFu;
    bool tab32 = false;
    for (int i = tid; i < T; i += NT) tab[i] = EMPTY64;
    if (tid < 32) sh[tid] = 0;
    if (tid < 16) ph[tid] = 0;
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    // the candidate's value is a function of its raw dot and of row constants only (see emit_candidates)
    const bool simple_judge = p.filter_mode != SP_SEL_MATRIX && p.target_mode != SP_SEL_MATRIX && p.l1 == 0.f && !(p.dbg & 256) &&
                              (p.fold || (p.l2 == 0.f && p.l3 == 0.f));
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
    // The current segment is the WAVE's (end of segment, flat->m2 index delta, m1 value in scalar registers, see below):
    // the common step costs one scalar compare and one add.  m2 is addressed with 32-bit byte offsets from the
    // scalar base pointers (64-bit ones in the BIG variant, which the host launches for nnz(m2) >= 2^30).
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
        // The segment of an element.  A wave's trip covers 64 CONSECUTIVE flat elements per unrolled step, so the walk along the segment
        // list is the WAVE's, not the lane's (round 6): `sc` is the last segment reached, `b` its end, {cur_delta, cur_v} its flat->m2 index
        // delta and m1 value — all wave-uniform, read from LDS at one address for the whole wave and kept in scalar registers; a boundary
        // inside the step's 64 elements hands the lanes behind it to the next segment (one compare, two selects).  Until round 6 every LANE
        // kept its own segment and walked on its own (exec-masked loops of dependent LDS gathers: with ~130-element slices and a stride of
        // 64 some lane crossed in nearly every step — more LDS instructions than the accumulation itself, 4 dependent round trips a step).
        auto rfl = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
        int sc = 0;
        {
            int sr = nb;      // last s in [0, nb) with seg_pre[s] <= e0 (seg_pre[0] = 0): uniform reads
            while (sr - sc > 1) {
                const int mid = (sc + sr) >> 1;
                if (rfl(seg_pre[mid]) <= e0) sc = mid; else sr = mid;
            }
        }
        int b = (sc + 1 < nb) ? rfl(seg_pre[sc + 1]) : 0x7FFFFFFF;      // first flat index beyond segment sc
        int cur_delta = rfl(seg_lo[sc] - seg_pre[sc]);                   // m2 position = flat index + delta
        float cur_v = __uint_as_float((unsigned)rfl((int)__float_as_uint(seg_v1[sc])));
        const int idx_safe = e0 + cur_delta;                             // (element e0 is real)
        auto fetch = [&](int ebase, int (&c)[AU], float (&x)[AU], float (&v1)[AU], unsigned &valid) {
            typename std::conditional<BIG, unsigned long long, unsigned>::type off[AU];
            valid = 0;
#pragma unroll
            for (int j = 0; j < AU; ++j) {
                const int rs = ebase + 64 * j;             // uniform
                const int ej = rs + lane;
                const bool ok = ej < e1;
                const int last = min(rs + 63, e1 - 1);     // uniform
                int my_delta = cur_delta;
                float my_v = cur_v;
                while (b <= last) {                        // uniform: a boundary at or before the step's last element (skips empty segments)
                    ++sc;
                    const int d_ = seg_lo[sc] - seg_pre[sc];
                    const float v_ = seg_v1[sc];
                    const int nxt = (sc + 1 < nb) ? seg_pre[sc + 1] : 0x7FFFFFFF;
                    cur_delta = rfl(d_);
                    cur_v = __uint_as_float((unsigned)rfl((int)__float_as_uint(v_)));
                    if (ej >= b) { my_delta = cur_delta; my_v = cur_v; }
                    b = rfl(nxt);
                }
                off[j] = (decltype(off[0] + 0))(unsigned)(ok ? ej + my_delta : idx_safe) << 2;
                v1[j] = ok ? my_v : 0.f;
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

    // rows of this kernel: descriptors [0, n_rows) of the generic queue (filled by the classification prepass and
    // by the sparse kernel's give-ups, both complete before this launch starts)
    const int n_rows = (int)p.qcount[1];
    const bool have_splits = p.splits != nullptr && (p.splits_state == nullptr || __builtin_amdgcn_readfirstlane(p.splits_state[0]) != 0);      // (uniform; written by an earlier launch)
    const int4 *desc = p.desc_g;
    int qi = 0;
    if (tid == 0) sh[SH_QA] = p.static_sched ? (int)blockIdx.x : (int)atomicAdd(&p.queue[1], 1u);
    __syncthreads();

    for (;;) {
        qi = __builtin_amdgcn_readfirstlane(sh[SH_QA]);
        if (qi >= n_rows) break;
        const int4 dC = desc[2 * (size_t)qi], wC = desc[2 * (size_t)qi + 1];
        // row-constant values are wave-uniform: v_readfirstlane moves them to scalar registers
        const int slot_i = __builtin_amdgcn_readfirstlane(dC.x);
        // a piece of a heavy row: a range of its fine column windows, results to the piece's own slot (sp_merge_pieces_kernel)
        const int piece = slot_i < 0 ? -1 - slot_i : -1;
        const int piece_range = piece >= 0 ? __builtin_amdgcn_readfirstlane(p.piece_info[piece].y) : 0;
        const int t = __builtin_amdgcn_readfirstlane(dC.y);
        const int s1 = __builtin_amdgcn_readfirstlane(dC.z);
        const int n1 = __builtin_amdgcn_readfirstlane(dC.w);
        const u64 macs = (u64)(unsigned)__builtin_amdgcn_readfirstlane(wC.x);   // saturated at 2^32-1 by the work prepass
        // claim the next queue position early; it is published at the bottom of the loop
        int next_q = 0;
        if (tid == 0) next_q = p.static_sched ? qi + (int)gridDim.x : (int)atomicAdd(&p.queue[1], 1u);

        RowCtx rc;
        rc.row = t;
        rc.have_thr = false;
        rc.thr_key = 0;
        Epi &epi = rc.epi;
        epi.a1 = p.a1; epi.l1 = p.l1; epi.l2 = p.l2; epi.l3 = p.l3; epi.t1 = p.t1; epi.t2 = p.t2;
        epi.stab = p.stab; epi.bayes = p.bayes; epi.threshold = p.threshold; epi.any = any_norm;
        epi.xtv = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(wC.y));    // row terms travel in the descriptor
        epi.xcos = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(wC.z));
        epi.xdep = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(wC.w));
        // den = l1*(t1*(X-xy) + t2*(Y-xy) + xy) + l2*Xc*Yc + l3*Xd*Yd + stab  >=  bA + bB*xy  when the
        // column terms are replaced by their minima and their multipliers are non-negative
        epi.bound = p.bound_ok && !(epi.xcos < 0.f) && !(epi.xdep < 0.f);
            epi.cut_ok = !((p.bayes != 0.f || p.l1 * (1.f - p.t1 - p.t2) > 0.f) && p.neg_flag != nullptr && *p.neg_flag != 0);
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
        // Selection on the candidate buffer.  Small LDS buffers (what every k up to ~800 gets) take the register-resident
        // radix selection of the sparse kernel: its four histograms live in seg_pre / seg_v1, which are dead from the end of the
        // accumulate to the next window's segment scan.  exact = false: a conservative cutoff after two digits is enough mid-row.
        auto select_now = [&](bool exact) __attribute__((always_inline)) -> long long {
            if constexpr (U_LDS && NT >= 512) {
                if (p.cap <= 2 * NT && !(p.dbg & 512)) {
                    int *hist4 = seg_pre;
                    for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
                    __syncthreads();
                    return select_fast<NT, false, 2>(U, hist4, sh, p.k, exact, rc.have_thr ? rc.thr_key : 0u);
                }
            }
            return compact_topk<NT>(U, hist, sh, p.k);
        };
        const bool row_done = (macs == 0);
        PHASE_END(PH_SETUP);

        // =========================================================================================
        // GENERIC path: accumulator tile + column windows
        // =========================================================================================
        if (!row_done) {
            bool retry_window = false;  // the current window repeats the previous lo (after an overflow)

            // dense windows can never overflow (one slot per column); hash windows are sized from the
            // MACs bound and split on overflow.  Window width w; windows are [lo, lo+w).
            // (All of this is 32-bit arithmetic with shifts for the powers of two — T, the fine window width — since round 6: the 64-bit divisions
            // and remainders it used to be are ~100 VALU instructions each, every wave computes them, and with four waves on a SIMD the
            // dozen of them per row and window were ~5 k cycles of every window: most of what the phase counter calls "segments".)
            const int Td = 2 * T;               // columns of a dense window (32-bit sums)
            int width;
            if (p.n_cols <= Td) {
                width = p.n_cols;
            } else {
                const unsigned p_dense = ((unsigned)p.n_cols + (unsigned)Td - 1u) >> (p.logT + 1);
                const unsigned m32 = (unsigned)macs, hf = (unsigned)max(1, p.hash_fill);      // (macs: saturated at 2^32 - 1 by the work prepass)
                const unsigned q = m32 / hf;
                const unsigned p_hash = q + ((m32 - q * hf) != 0u ? 1u : 0u);
                if (p_hash < 1u || p_dense <= p_hash) width = Td;
                else width = (int)(((unsigned)p.n_cols + p_hash - 1u) / p_hash);
            }
            const int split_shift = 31 - __builtin_clz((unsigned)max(1, p.split_w));      // (the fine window width is a power of two: 2T / f, make_config)
            const int split_mask = (1 << split_shift) - 1;

            int lo = 0, col_end = p.n_cols;
            bool first_window = true;      // of this row or piece (uniform)
            if (piece >= 0) {       // (the splitter only cuts rows of a call with standard windows: n_cols > Td)
                width = Td;
                lo = (int)min((long long)p.n_cols, (long long)(piece_range & 0xFFFF) << split_shift);
                col_end = (int)min((long long)p.n_cols, (long long)(piece_range >> 16) << split_shift);
            }
            // A light row's NEXT window (round 6): the m1 row fits one batch, so thread i keeps segment i's m2 row, m1 value and this window's
            // end — the next window's start — in registers and asks for the next window's end (one gather from the boundary table) while
            // this window is accumulated and drained: the window after the first starts without a load (the two dependent round trips, m1
            // entry -> boundary positions, were 1.3 k + 3.7 k cycles of every window, and the waves' skew on them most of the scan's barrier).
            int pf_u = 0, pf_r0 = 0, pf_r1 = 0;
            float pf_v = 0.f;
            bool pf_have = false;          // (uniform) the registers hold the window that comes now
            while (lo < col_end) {
                const int hi = (col_end - lo <= width) ? col_end : lo + width;
                const int wlo = lo, whi = hi;
                const bool dense = (hi - lo) <= Td;
                const bool whole = (wlo == 0 && whi == p.n_cols);
                if (dense != tab32) {           // (uniform; the previous drain ended with a barrier)
                    if (dense) { for (int i = tid; i < 2 * T; i += NT) tabw[i] = EMPTY32; }
                    else       { for (int i = tid; i < T; i += NT) tab[i] = EMPTY64; }
                    tab32 = dense;
                    __syncthreads();
                }
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
                const bool use_splits = have_splits && width == Td && (lo & split_mask) == 0 && (hi == p.n_cols || (hi & split_mask) == 0);   // (a hashed row halved down to Td may sit elsewhere)
                // A row of SEVERAL batches (more m1 entries than threads: the heavy rows and their pieces) on the boundary table: the NEXT batch's
                // entry, slice bounds and m1 value are requested while this batch is scanned and accumulated (in the registers a light row keeps its
                // next window in: a row is one or the other) — the two dependent round trips in front of every batch of ~14 k products were a
                // seventh of a heavy row's time.
                const bool bpf = !carry && (use_splits || whole) && !(p.dbg & 4194304);
                auto batch_request = [&](int bq) __attribute__((always_inline)) {
                    if (tid < min(NT, n1 - bq)) {
                        const int u = p.m1_indices[s1 + bq + tid];
                        pf_r0 = (wlo != 0) ? p.splits[(size_t)((wlo >> split_shift) - 1) * (size_t)p.splits_rows + (size_t)u] : p.m2_indptr[u];
                        pf_r1 = (whi < p.n_cols) ? p.splits[(size_t)((whi >> split_shift) - 1) * (size_t)p.splits_rows + (size_t)u] : p.m2_indptr[u + 1];
                        pf_v = p.m1_data[s1 + bq + tid];
                    }
                };
                if (bpf) batch_request(0);
                for (int b0 = 0; b0 < n1; b0 += NT) {
                    const int nb = min(NT, n1 - b0);
                    int len = 0;
                    if (bpf) {
                        if (tid < nb) {
                            seg_lo[tid] = pf_r0;
                            seg_v1[tid] = pf_v;
                            len = pf_r1 - pf_r0;
                        }
                        if (b0 + NT < n1) batch_request(b0 + NT);
                    } else {
                    // the window after this one, when it is a standard dense window of the boundary table too (dense windows never overflow:
                    // the loop below arrives at exactly these bounds)
                    const int nhi = (col_end - hi <= width) ? col_end : hi + width;
                    const bool pf_next = carry && use_splits && dense && hi < col_end && (hi & split_mask) == 0 && (nhi == p.n_cols || (nhi & split_mask) == 0) &&
                                         !(p.dbg & 4194304);      // (bit 4194304 of the ablation word: off, for A/B runs)
                    if (pf_have) {
                        if (tid < nb) {
                            seg_lo[tid] = pf_r0;
                            seg_v1[tid] = pf_v;
                            len = pf_r1 - pf_r0;
                            pf_r0 = pf_r1;
                            if (pf_next) pf_r1 = (nhi < p.n_cols) ? p.splits[(size_t)((nhi >> split_shift) - 1) * (size_t)p.splits_rows + (size_t)pf_u] : p.m2_indptr[pf_u + 1];
                        }
                    } else
                    if (tid < nb) {
                        const int u = p.m1_indices[s1 + b0 + tid];
                        int r0 = p.m2_indptr[u], r1 = p.m2_indptr[u + 1];
                        if (pf_next) {
                            pf_u = u;
                            pf_r1 = (nhi < p.n_cols) ? p.splits[(size_t)((nhi >> split_shift) - 1) * (size_t)p.splits_rows + (size_t)u] : r1;
                        }
                        if (!whole && use_splits) {
                            // both ends are multiples of the fine window width: their positions were found once per call (sp_m2_splits_kernel)
                            if (wlo != 0) r0 = p.splits[(size_t)((wlo >> split_shift) - 1) * (size_t)p.splits_rows + (size_t)u];
                            if (whi < p.n_cols) r1 = p.splits[(size_t)((whi >> split_shift) - 1) * (size_t)p.splits_rows + (size_t)u];
                        } else if (!whole) {
                            // slice of the sorted m2 row inside [wlo, whi)  (s_plus.h:385-394)
                            if (wlo != 0) {
                                if (carry && !first_window) r0 = retry_window ? seg_lo[tid] : seg_hi[tid];      // (a piece's first window starts inside the row: nothing to chain to)
                                else r0 = lower_bound_g(p.m2_indices, r0, r1, wlo);
                            }
                            if (whi < p.n_cols) r1 = lower_bound_g(p.m2_indices, r0, r1, whi);
                            if (carry) seg_hi[tid] = r1;
                        }
                        seg_lo[tid] = r0;
                        const float v = p.m1_data[s1 + b0 + tid];
                        seg_v1[tid] = v;
                        len = r1 - r0;
                        pf_r0 = r1; pf_v = v;
                    }
                    pf_have = pf_next;
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
                            // direct-indexed window: every column owns its 32-bit sum.  Optimistic update: read the
                            // slot, then ONE compare-and-swap writes sum + x (ds_cmpst_rtn_b32: 3-6 lanes/clk against
                            // 0.33 for ds_add_f32); the lanes of a wave instruction hold 64 distinct columns of one
                            // m2 row, so only another wave can interfere — a lost race falls back to the hardware
                            // float add (whoever won left a real sum there), which cannot livelock on hot columns.
                            unsigned cur[ACC_UNROLL], prev[ACC_UNROLL];
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) cur[j] = tabw[c[j] - wlo];
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j) {
                                const float sum = (cur[j] == EMPTY32 ? 0.f : __uint_as_float(cur[j])) + x[j];
                                prev[j] = atomicCAS(&tabw[c[j] - wlo], cur[j], __float_as_uint(sum));
                            }
#pragma unroll
                            for (int j = 0; j < ACC_UNROLL; ++j)
                                if (prev[j] != cur[j]) atomicAdd((float *)&tabw[c[j] - wlo], x[j]);
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
                        width = max(T, (width >> 1) + (width & 1));
                        retry_window = true;  // same lo again: slice starts are still in seg_lo
                        __syncthreads();
                        continue;
                    }
                }
                retry_window = false;
                if (timing) ph[CT_PASSES] += 1;

                // ---- a cutoff BEFORE the drain, without a selection (round 5; dense windows, value a growing function of the raw dot alone).
                // The drain used to start without any k-th value: the first 1 824 touched slots filled the candidate buffer, a selection gave the
                // k-th of those (the top ~11 %), the sweep went on letting 11 % through, filled the buffer again ... 2.8 selections and 1.8
                // sweeps per row on the MovieLens shape.  Now every thread takes the largest raw dot of its slots, every wave the
                // ceil(k / NW)-th largest of its lanes' maxima (as the sparse kernel's first stage does): the minimum over the waves is a raw
                // dot that at least k columns of THIS window reach — when every wave had that many lanes with a live slot; else nothing
                // changes.  One more pass over the window's sums in LDS, no pushes, two barriers. ----
                if (dense && simple_judge && p.a1 == 1.f && p.bayes == 0.f && U_LDS && (p.k + NW - 1) / NW <= 32 && !(p.dbg & 131072) && (!rc.have_thr || (p.dbg & 2097152))) {      // (bit 131072 of the ablation word: off, for A/B runs)
                    const float slope = any_norm ? rc.epi(1.f, 0.f, 1.f, 1.f) : 1.f;      // uniform: val = slope * xy
                    if (slope > 0.f && slope < __builtin_inff()) {      // uniform
                        unsigned lmax = 0u;
                        // (four consecutive sums per 16-byte LDS read; slots behind the window's end hold EMPTY32)
                        for (int s4 = 4 * tid; s4 < t_eff && s4 + 3 < 2 * T; s4 += 4 * NT) {
                            const uint4 w4 = *(const uint4 *)&tabw[s4];
                            const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (w[j] != EMPTY32 && __uint_as_float(w[j]) > rc.xy_cut) lmax = max(lmax, fkey(__uint_as_float(w[j])));      // (a NaN sum compares false: no part of the statistic)
                        }
                        if (tid == 0) { sh[SH_SEL] = -1; sh[SH_NEED] = 1; }
                        __syncthreads();
                        // the wave's ceil(k / NW)-th largest lane maximum, from below: the largest T (of the keys' upper 24 bits) that so many
                        // lanes reach — a bit-wise search, one compare and a scalar count per bit (round 6; it was `rounds` extractions of the
                        // wave's maximum, ~220 vector instructions on a SIMD that four waves share); 0 when the wave has too few live lanes
                        const int rounds = (p.k + NW - 1) / NW;
                        unsigned tw = 0u;
                        for (int bit = 31; bit >= 8; --bit) {
                            const unsigned cand = tw | (1u << bit);
                            if (__popcll(__ballot(lmax >= cand)) >= rounds) tw = cand;      // uniform
                        }
                        if (lane == 0) { if (tw != 0u) atomicMin((unsigned *)&sh[SH_SEL], tw); else sh[SH_NEED] = 0; }
                        __syncthreads();
                        const unsigned g = (unsigned)sh[SH_SEL];
                        const bool ok = sh[SH_NEED] != 0 && g != 0xFFFFFFFFu;
                        __syncthreads();      // (both words are the selections' too)
                        if (ok) {
                            // at least k columns of this window have a raw dot >= funkey(g), i.e. a value >= val(g): everything that cannot beat the
                            // key just below val(g) is out (the cutoff is strict; the columns AT val(g) are among the k)
                            const float vg = any_norm ? rc.epi(funkey(g), 0.f, 1.f, 1.f) : funkey(g);
                            if (vg == vg) {
                                const unsigned kg = fkey(vg);
                                const unsigned below = kg > 0u ? kg - 1u : 0u;
                                if (!rc.have_thr || below > rc.thr_key) {
                                    rc.have_thr = true;
                                    rc.thr_key = below;
                                    rc.set_cut(p.threshold);
                                }
                            }
                        }
                    }
                }

                // ================= drain: one barrier-free sweep, overflow-retry =================
                // Dense windows under the simple judge (round 5): with a cutoff in place ~250 of a window's 32 768 sums are live, one or two per
                // wave and trip — and every trip that held ONE paid the whole judge (epilogue, key, ballots, the reservation's atomic and its
                // round trip): 8.8 of the drain's 19 ms on the MovieLens shape (ablations in profiles/r05_exp_dropped.txt).  The sweep now only
                // SORTS: four consecutive sums per 16-byte read, dead ones cleared by one 16-byte write, the slot numbers of live ones appended
                // to a wave-private list (no atomics: the counter is a scalar register; 128 entries per wave in the dead seg_v1 array), and
                // the list is judged densely — 64 entries per trip — when it fills and at the sweep's end.  A trip that is mostly live (no
                // cutoff yet) is judged on the spot as before.
                // (general epilogues too: the sweep's test is then the gather-free upper bound — candidate_live — and the dense pass over the
                // list does the gathers, the selectors and the epilogue)
                const bool fast_drain = dense && !(p.dbg & 262144);      // (bit 262144 of the ablation word: off)
                for (;;) {
                    if (fast_drain) {
                        // (two instantiations: the simple judge's sweep is one compare per sum and must not carry the general test's code)
                        auto fast_sweep = [&](auto simple_c) __attribute__((always_inline)) {
                        constexpr bool SIMPLE = decltype(simple_c)::value;
                        constexpr int WLCAP = 2 * NT / NW;                      // u16 entries per wave
                        unsigned short *wl = (unsigned short *)seg_v1 + wave * WLCAP;
                        int wn = 0;                                             // wave-uniform
                        auto flush = [&]() __attribute__((always_inline)) {
                            for (int b = 0; b < wn; b += 64) {
                                if (sh[SH_RETRY]) break;                        // (U is full: what is left stays in the tile for the sweep after the selection)
                                const int i = b + lane;
                                if constexpr (SIMPLE) {
                                    // an entry is a GROUP of four consecutive sums, dead ones already cleared by the sweep
                                    int c[4];
                                    float xy[4];
                                    unsigned occ = 0;
                                    int g4 = -1;
                                    unsigned w[4] = {EMPTY32, EMPTY32, EMPTY32, EMPTY32};
                                    if (i < wn) {
                                        g4 = 4 * (int)wl[i];
                                        const uint4 w4 = *(const uint4 *)&tabw[g4];
                                        w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
                                    }
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        c[j] = (w[j] != EMPTY32) ? wlo + g4 + j : EMPTY;
                                        xy[j] = (w[j] != EMPTY32) ? __uint_as_float(w[j]) : 0.f;
                                        if (w[j] != EMPTY32) occ |= 1u << j;
                                    }
                                    const unsigned done = emit_candidates<4>(p, rc, c, xy, occ, U, sh, p.cap, true);
                                    if (occ) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) if (done & (1u << j)) w[j] = EMPTY32;
                                        *(uint4 *)&tabw[g4] = make_uint4(w[0], w[1], w[2], w[3]);
                                    }
                                } else {
                                int c[1];
                                float xy[1];
                                unsigned occ = 0;
                                int sidx = -1;
                                c[0] = EMPTY; xy[0] = 0.f;
                                if (i < wn) {
                                    sidx = (int)wl[i];
                                    const unsigned w = tabw[sidx];
                                    if (w != EMPTY32) { c[0] = wlo + sidx; xy[0] = __uint_as_float(w); occ = 1u; }
                                }
                                const unsigned done = emit_candidates<1>(p, rc, c, xy, occ, U, sh, p.cap, SIMPLE);
                                if (done & 1u) tabw[sidx] = EMPTY32;
                                }
                            }
                            wn = 0;
                        };
                        // one step of the sweep: a thread's four consecutive sums (already read)
                        auto step = [&](int s4, bool in, const uint4 &w4) __attribute__((always_inline)) {
                            unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
                            if constexpr (SIMPLE) {
                                // (round 6) one compare per sum — an untouched slot's pattern and a NaN sum compare false (the exact test would drop
                                // the NaN anyway) —, ONE ballot per trip: the lanes that hold a live sum append their GROUP of four to the wave's
                                // list, everything else of the group is cleared by one 16-byte write.  More than half of the lanes live (no cutoff
                                // yet): judged on the spot, below.
                                bool lv[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) lv[j] = __uint_as_float(w[j]) > rc.xy_cut;
                                const bool any = (lv[0] | lv[1]) | (lv[2] | lv[3]);
                                const u64 ma = __ballot(any);
                                const int na = __popcll(ma);
                                if (na <= 32) {
                                    if (in) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) w[j] = lv[j] ? w[j] : EMPTY32;
                                        *(uint4 *)&tabw[s4] = make_uint4(w[0], w[1], w[2], w[3]);
                                    }
                                    if (na) {
                                        if (wn + na > WLCAP) flush();
                                        if (any) wl[wn + mbcnt64(ma)] = (unsigned short)(s4 >> 2);
                                        wn += na;
                                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                    }
                                    return;
                                }
                            }
                            bool live[4];
                            bool any_dead = false;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                live[j] = w[j] != EMPTY32 && (SIMPLE ? !(__uint_as_float(w[j]) <= rc.xy_cut)      // (a NaN sum stays live: the exact test drops it)
                                                                           : candidate_live(p, rc, __uint_as_float(w[j])));
                                any_dead |= (w[j] != EMPTY32) && !live[j];
                            }
                            const u64 m0 = __ballot(live[0]), m1 = __ballot(live[1]), m2 = __ballot(live[2]), m3 = __ballot(live[3]);
                            const int tot = (__popcll(m0) + __popcll(m1)) + (__popcll(m2) + __popcll(m3));
                            if (SIMPLE || tot > WLCAP / 2) {
                                // mostly live (no cutoff yet): judged on the spot
                                int c[4];
                                float xy[4];
                                unsigned occ = 0;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    c[j] = (w[j] != EMPTY32) ? wlo + s4 + j : EMPTY;
                                    xy[j] = (w[j] != EMPTY32) ? __uint_as_float(w[j]) : 0.f;
                                    if (w[j] != EMPTY32) occ |= 1u << j;
                                }
                                const unsigned done = emit_candidates<4>(p, rc, c, xy, occ, U, sh, p.cap, SIMPLE);
#pragma unroll
                                for (int j = 0; j < 4; ++j) if (done & (1u << j)) w[j] = EMPTY32;
                                if (in && occ) *(uint4 *)&tabw[s4] = make_uint4(w[0], w[1], w[2], w[3]);
                                return;
                            }
                            if (any_dead) {      // the dead ones are consumed here: one 16-byte write (live sums are written back as they are)
#pragma unroll
                                for (int j = 0; j < 4; ++j) if (!live[j]) w[j] = EMPTY32;
                                *(uint4 *)&tabw[s4] = make_uint4(w[0], w[1], w[2], w[3]);
                            }
                            if (tot) {
                                if (wn + tot > WLCAP) flush();
                                if (live[0]) wl[wn + mbcnt64(m0)] = (unsigned short)(s4);
                                wn += __popcll(m0);
                                if (live[1]) wl[wn + mbcnt64(m1)] = (unsigned short)(s4 + 1);
                                wn += __popcll(m1);
                                if (live[2]) wl[wn + mbcnt64(m2)] = (unsigned short)(s4 + 2);
                                wn += __popcll(m2);
                                if (live[3]) wl[wn + mbcnt64(m3)] = (unsigned short)(s4 + 3);
                                wn += __popcll(m3);
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            }
                        };
                        for (int base = 0; base < t_eff; base += 4 * NT) {
                            if (sh[SH_RETRY]) break;         // (one LDS word, the same for every lane: wave-uniform)
                            const int s4 = base + 4 * tid;
                            const bool in = s4 < t_eff && s4 + 3 < 2 * T;
                            uint4 w4 = make_uint4(EMPTY32, EMPTY32, EMPTY32, EMPTY32);
                            if (in) w4 = *(const uint4 *)&tabw[s4];
                            step(s4, in, w4);
                        }
                        if (wn) flush();
                        };
                        if (simple_judge) fast_sweep(std::true_type{}); else fast_sweep(std::false_type{});
                    } else
                    for (int base = 0; base < t_eff; base += NT * DRAIN_UNROLL) {
                        // the candidate buffer is full: whatever is judged now cannot be stored, and it would be judged
                        // without the k-th value the selection is about to give — stop, select, sweep again (the
                        // slots already consumed are empty and cost a read)
                        if (sh[SH_RETRY]) break;         // (one LDS word, the same for every lane: wave-uniform)
                        int c[DRAIN_UNROLL];
                        float xy[DRAIN_UNROLL];
                        unsigned occ = 0;
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j) {
                            const int sidx = base + j * NT + tid;
                            c[j] = EMPTY;
                            xy[j] = 0.f;
                            if (sidx < t_eff) {
                                if (dense) {
                                    const unsigned w = tabw[sidx];
                                    if (w != EMPTY32) { c[j] = wlo + sidx; xy[j] = __uint_as_float(w); }
                                } else {
                                    const u64 slot = tab[sidx];
                                    c[j] = (int)(slot >> 32);
                                    xy[j] = __uint_as_float((unsigned)slot);
                                }
                            }
                            if (c[j] != EMPTY) occ |= 1u << j;
                        }
                        const unsigned done = (p.dbg & 32) ? occ : emit_candidates<DRAIN_UNROLL>(p, rc, c, xy, occ, U, sh, p.cap, simple_judge);   // (ablation: scan and clear only)
#pragma unroll
                        for (int j = 0; j < DRAIN_UNROLL; ++j)
                            if (done & (1u << j)) {
                                if (dense) tabw[base + j * NT + tid] = EMPTY32;
                                else tab[base + j * NT + tid] = EMPTY64;
                            }
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
                    if (timing) ph[8] += 1;          // event count: sweeps repeated after a full candidate buffer
                    took_threshold(select_now(false));
                    PHASE_END(PH_SELECT);
                }
                PHASE_END(PH_DRAIN);
                lo = hi;
                first_window = false;
            }
        }


        // ================= final selection + write-out =================
        __syncthreads();
        const int n_fin = sh[SH_CNT];
        __syncthreads();
        if (n_fin > p.k) took_threshold(select_now(true));
        PHASE_END(PH_SELECT);
        const int n_out = sh[SH_CNT];
        if (piece >= 0) {
            const long long o = (long long)piece * (long long)p.k;
            for (int j = tid; j < n_out; j += NT) {
                const u64 it = U[j];
                p.part_cols[o + j] = (int)(unsigned)(it & 0xFFFFFFFFull);
                p.part_vals[o + j] = funkey((unsigned)(it >> 32));
            }
            if (tid == 0) {
                p.part_counts[piece] = n_out;
                sh[SH_QA] = next_q;
            }
        } else {
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
            sh[SH_QA] = next_q;
        }
        }
        __syncthreads();
        if (tid == 0) sh[SH_CNT] = 0;
        __syncthreads();
        PHASE_END(PH_OUTPUT);
    }
    if (timing) {
#pragma unroll
        for (int i = 0; i < PH_N; ++i) atomicAdd(&p.phase_cycles[i], ph[i]);
    }
#undef PHASE_END
}

}  // namespace
