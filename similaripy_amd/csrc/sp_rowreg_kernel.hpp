// sp_rowreg_kernel.hpp — the headline shape (BASELINE configs[1]) of the bitmap + two-sweeps row kernel of
// sp_sparse_kernel.hpp, restructured around what bounds that kernel on gfx950: a CU pulls ~13 B/clk of HBM stream however
// many waves ask, and the memory pipe sits idle during the dense phases of a row (42 % of its time).
//   * A row's work items (<= 256 consecutive elements of one m2 row, heaviest segments first) are cut ONCE PER CALL by
//     sp_row_items_kernel into global memory, in wave-major order: the row kernel has no setup phase, a wave fetches its
//     <= RR_NSLOT item descriptors with one 16-byte load per lane, a row ahead.
//   * The column ids of a wave's items stay in VGPRs (RR_NSLOT x 4 per lane) through both sweeps: sweep 2 streams the
//     values only — 8 B per MAC from memory instead of 12 (s_plus.h:411-441 reads each product once, too).
//   * Those id loads are ISSUED when the previous row's last sweep ends, all of them at once: they land while the
//     previous row accumulates its collision set, selects and writes out, so sweep 1 is LDS work on resident registers.
//     The workgroup barriers in between wait for LDS traffic only (wg_sync<true>), never for the loads in flight.
// Monotone epilogues only (val = xy / den or the raw dot, see sp_sparse_kernel.hpp), 1024 threads, candidate buffer in
// LDS, rows of <= 64 m1 entries and 16 .. RR_ICAP items; every other row is classified for the other two row kernels.
// A row whose pools overflow is handed to the generic kernel's queue, as there.
#pragma once
#include "sp_common.hpp"

namespace {

constexpr int RR_NT = 1024;
constexpr int RR_NW = RR_NT / 64;
#ifndef RR_NSLOT_D
#define RR_NSLOT_D 14
#endif
#ifndef RR_PF_D
#define RR_PF_D 3
#endif
constexpr int RR_NSLOT = RR_NSLOT_D;             // items per wave held in registers (4 VGPRs of column ids each)
constexpr int RR_ICAP = RR_NW * RR_NSLOT;      // items per row
constexpr int RR_STRIDE = RR_ICAP + 1;         // int4 records per row in items_g: header, then wave w's slots at 1 + w*RR_NSLOT
constexpr int RR_PF = RR_PF_D;                       // value loads of sweep 2 in flight per wave
constexpr int RR_KMAX = 14 * RR_NW;            // the selection-free first stage needs (k + NW - 1) / NW + 2 <= 16 rounds

// ---- work items of the rows in the register-resident queue, once per call.  One wave per row: segment i = m1 entry i of
// the row, visited in descending |m1 value| (each segment scales its m2 row by its m1 value: the large products come first
// and the running k-th value — the cutoff of everything after — starts high), cut into items of <= ITEM elements.
// Item n of the row is slot n / 16 of wave n % 16; unused slots hold the all-out-of-range sentinel. ----
__global__ __launch_bounds__(256) void sp_row_items_kernel(const unsigned *__restrict__ rq, int rr_cap, const int4 *__restrict__ desc_r,
                                                            const int *__restrict__ m1_indices, const float *__restrict__ m1_data,
                                                            const int *__restrict__ m2_indptr, int4 *__restrict__ items_g) {
    const int lane = threadIdx.x & 63;
    const int n_rows = (int)min(rq[1], (unsigned)rr_cap);
    const int waves_total = (int)(gridDim.x * (blockDim.x >> 6));
    for (int q = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); q < n_rows; q += waves_total) {
        const int4 d = desc_r[2 * (size_t)q];
        const int s = __builtin_amdgcn_readfirstlane(d.z), n1 = __builtin_amdgcn_readfirstlane(d.w);      // n1 <= 64 (classification)
        int r0 = 0, len = 0;
        unsigned vbits = 0u;
        if (lane < n1) {
            const int u = m1_indices[s + lane];
            vbits = __float_as_uint(m1_data[s + lane]);
            r0 = m2_indptr[u];
            len = m2_indptr[u + 1] - r0;
        }
        const unsigned key = (lane < n1 && len > 0) ? ((vbits & 0x7FFFFFFFu) | 1u) : 0u;      // 0 = no segment
        // position of this lane's segment in descending key order (ties: lower lane first; empty lanes last): a permutation
        int rank = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, j);
            rank += (kj > key || (kj == key && j < lane)) ? 1 : 0;
        }
        const int nit = (len + ITEM - 1) / ITEM;
        // values in position order (lane i sends to lane rank_i), scanned there, and read back
        const int nit_p = __builtin_amdgcn_ds_permute(rank * 4, key != 0u ? nit : 0);
        const int len_p = __builtin_amdgcn_ds_permute(rank * 4, key != 0u ? len : 0);
        const int ib_incl = wave_incl_scan_dpp(nit_p), fs_incl = wave_incl_scan_dpp(len_p);
        const int ib = __builtin_amdgcn_ds_bpermute(rank * 4, ib_incl - nit_p);
        const int fs = __builtin_amdgcn_ds_bpermute(rank * 4, fs_incl - len_p);
        const int n_items = __builtin_amdgcn_readlane(ib_incl, 63);
        int4 *row = items_g + (size_t)q * RR_STRIDE;
        // sentinel in every slot the row does not use (n_items <= RR_ICAP by classification)
        for (int i = n_items + lane; i < RR_ICAP; i += 64)
            row[1 + (i % RR_NW) * RR_NSLOT + i / RR_NW] = make_int4((int)OOB_SOFFSET, 0, 0, 0);
        int l16 = 0;      // lanes (4 elements each) of the row's first 16 items: what the selection-free first stage looks at
        if (key != 0u) {
            int n = ib;
            for (int o = 0; o < len; o += ITEM, ++n) {
                const int cnt = min(ITEM, len - o);
                if (n < RR_ICAP) row[1 + (n % RR_NW) * RR_NSLOT + n / RR_NW] = make_int4((r0 + o) * 4, cnt, (int)vbits, fs + o);
                if (n < RR_NW) l16 += (cnt + 3) / 4;
            }
        }
        l16 = wave_incl_scan_dpp(l16);
        if (lane == 63) row[0] = make_int4(n_items, l16, 0, 0);
    }
}

__global__ __launch_bounds__(RR_NT) void sp_knn_rowreg_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = RR_NT, NW = RR_NW;
    int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A_bytes = p.T * 8;

    // ---- LDS carve-up: as in sp_sparse_kernel.hpp, without the item list ----
    // cbm[CBM_BYTES] collision bitmap | pre16[] its popcount prefix | region A: column bitmap during sweep 1, afterwards
    // [0,A/4) collision set, [A/2,3A/4) member pool, [3A/4,A) candidate buffer U | hist4[4][256] | sh[32] | ph[16]
    unsigned char *cbm = smem;
    unsigned short *pre16 = (unsigned short *)(smem + CBM_BYTES);
    unsigned char *rA = smem + CBM_BYTES + PRE_BYTES;
    int *hist4 = (int *)(rA + A_bytes);
    int *sh = hist4 + 1024;
    u64 *ph = (u64 *)(sh + 32);
    u64 *U = (u64 *)(rA + (A_bytes / 4) * 3);
    const int cap = p.cap_s;

    const unsigned amask = (unsigned)((1u << (p.nb_log2 - 3)) - 1u) & ~3u;
    const int nb_bytes = 1 << (p.nb_log2 - 3);
    const unsigned cmask = (unsigned)(CBM_BYTES - 1) & ~3u;
    u64 *cs = (u64 *)rA;
    const int CSN = A_bytes / 32;
    const int cs_shift = 32 - (p.logT - 2);
    u64 *mpool = (u64 *)(rA + A_bytes / 2);
    const int mpcap = A_bytes / 32;
    const unsigned mpool_off = (unsigned)(CBM_BYTES + PRE_BYTES + A_bytes / 2);
    const unsigned u_off = (unsigned)(CBM_BYTES + PRE_BYTES + (A_bytes / 4) * 3);
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_indices, 0, (int)p.m2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_val = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_data, 0, (int)p.m2_bytes, 0x00020000);

    for (int i = tid; i < (CBM_BYTES + PRE_BYTES + A_bytes) / 16; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
    for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
    if (tid < 32) sh[tid] = 0;
    if (tid < 16) ph[tid] = 0;
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    // raw dot: x <= cut_raw  =>  x < threshold (a scalar, computed once)
    const float cut_raw = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)RowCtx::funkey_inv_below(p.threshold)));
    const bool timing = (p.phase_cycles != nullptr) && tid == 0;
    u64 tmark = timing ? (u64)clock64() : 0;
#define PHASE_END(which) do { if (timing) { const u64 _n = (u64)clock64(); ph[which] += _n - tmark; tmark = _n; } } while (0)

    // ---- row pipeline: queue position -> descriptor -> item descriptors -> column ids, each a row ahead of its use ----
    const int n_rows = (int)min(p.rq[1], (unsigned)p.rr_cap);
    const int4 *desc = p.desc_r;
    // a row descriptor travels in ONE vector register: lane i < 8 holds dword i of the 32-byte record {slot, m1 row, m1 start,
    // m1 length | MACs, den, -, -}; slot = -1 beyond the queue's end
    auto load_desc = [&](int q) -> int {
        int w = (lane == 0) ? -1 : 0;
        if (q < n_rows && lane < 8) w = ((const int *)desc)[8 * (size_t)q + lane];
        return w;
    };
    // item descriptors of queue position q: slot `lane` of this wave; lane RR_NSLOT: the row's header {items, lanes of the
    // first 16 items}; the lanes beyond: the sentinel
    auto load_items = [&](int q, int &dx, int &dy, int &dz) {
        dx = (int)OOB_SOFFSET; dy = 0; dz = 0;
        if (q < n_rows && lane <= RR_NSLOT) {
            const int4 *row = p.items_g + (size_t)q * RR_STRIDE;
            const int4 d = row[(lane < RR_NSLOT) ? 1 + (tid >> 6) * RR_NSLOT + lane : 0];
            dx = d.x; dy = d.y; dz = d.z;
        }
    };
    u32x4 ids[RR_NSLOT];
    // one 16-byte load per lane fetches the column ids of a whole item; lanes beyond a partial item's end and every lane of an
    // unused slot get an out-of-range offset: they fetch nothing and read 0 (scripts/buffer_oob_probe.hip)
    auto issue_ids = [&](int dx, int dy) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RR_NSLOT; ++j) {
            const int off = __builtin_amdgcn_readlane(dx, j), cnt = __builtin_amdgcn_readlane(dy, j);
            const int vo = (4 * lane < cnt) ? lane * 16 : (int)(OOB_SOFFSET - (unsigned)off);
            ids[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, vo, off, 0);
        }
    };

    int q_c = 0, q_n = 0, q_nn = 0;      // queue positions of the current row and the two after it
    if (p.static_sched) {
        q_c = (int)blockIdx.x; q_n = q_c + (int)gridDim.x; q_nn = q_n + (int)gridDim.x;
    } else {
        if (tid == 0) {
            sh[SH_QA] = (int)atomicAdd(&p.rq[0], 1u);
            sh[SH_QB] = (int)atomicAdd(&p.rq[0], 1u);
            sh[SH_NITEMS] = (int)atomicAdd(&p.rq[0], 1u);      // (the position three rows ahead waits in LDS)
        }
        __syncthreads();
        q_c = sh[SH_QA]; q_n = sh[SH_QB];
        __syncthreads();
    }
    int dwC = load_desc(q_c), dwN = load_desc(q_n);
    int myd_x, myd_y, myd_z, mydN_x, mydN_y, mydN_z;
    load_items(q_c, myd_x, myd_y, myd_z);
    issue_ids(myd_x, myd_y);

    for (;;) {
        // (the thread id is made opaque once per row: everything derived from it — LDS addresses of a dozen loops — is then
        // recomputed where it is used instead of being hoisted out of the row loop into registers that live through it;
        // the column ids need those registers)
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        const int slot_i = __builtin_amdgcn_readlane(dwC, 0);
        if (slot_i < 0) break;
        const int t = __builtin_amdgcn_readlane(dwC, 1);
        const unsigned macs32 = (unsigned)__builtin_amdgcn_readlane(dwC, 4);
        const int n_items = __builtin_amdgcn_readlane(myd_x, RR_NSLOT);
        const int l16 = max(1, __builtin_amdgcn_readlane(myd_y, RR_NSLOT));
        const int n_slots = (n_items + NW - 1) / NW;

        // next row's item descriptors; queue position three rows ahead
        load_items(q_n, mydN_x, mydN_y, mydN_z);
        if (!p.static_sched && tid == 0) sh[SH_QA] = sh[SH_NITEMS];
        if (tid == 0) {
            int minus1;
            asm volatile("v_mov_b32 %0, -1" : "=v"(minus1));      // (a constant the compiler cannot park in a register across the row loop)
            sh[SH_PCTR] = 0; sh[SH_MCTR] = 0; sh[SH_CNT] = 0; sh[SH_SEL] = minus1; sh[SH_NEED] = 0;
        }
        bool failed = false;
        PHASE_END(PH_SETUP);

        float cutx = -__builtin_inff();    // a single product / a column sum <= cutx cannot enter the top-k
        float cutx0 = cutx;                // the part of it that comes from the `threshold` parameter
        const float den = __uint_as_float((unsigned)__builtin_amdgcn_readlane(dwC, 5));   // val = xy / den
        bool have_thr = false;
        unsigned thr_key = 0u;
        {
            if (!any_norm) {
                cutx0 = cut_raw;
            } else {
                const float c0 = p.threshold * den;
                cutx0 = c0 - fabsf(c0) * 2e-6f - 1e-37f;
                if (!(c0 == c0)) cutx0 = -__builtin_inff();
            }
            cutx = cutx0;
        }

        // ---- sweep 1 on the resident column ids: every product ORs its bit into the bitmap; the returned word tells whether
        // the column was there already, in which case its bit is ORed into the collision bitmap as well ----
#pragma unroll
        for (int j = 0; j < RR_NSLOT; j += 2) {
            const int cnt0 = __builtin_amdgcn_readlane(myd_y, j), cnt1 = __builtin_amdgcn_readlane(myd_y, j + 1);
            if (cnt0 != 0) {      // (wave-uniform; slots fill in order)
                const unsigned c[8] = {ids[j].x, ids[j].y, ids[j].z, ids[j].w, ids[j + 1].x, ids[j + 1].y, ids[j + 1].z, ids[j + 1].w};
                unsigned seen[8];
                if (cnt0 == ITEM && cnt1 == ITEM) s1_core8<CBM_BYTES + PRE_BYTES, false>(c, 4 * lane, cnt0, cnt1, amask, seen);
                else s1_core8<CBM_BYTES + PRE_BYTES, true>(c, 4 * lane, cnt0, cnt1, amask, seen);      // padding ORs nothing
                if (__ballot(((seen[0] | seen[1]) | (seen[2] | seen[3]) | (seen[4] | seen[5]) | (seen[6] | seen[7])) != 0u)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (seen[i]) atomicOr((unsigned *)(cbm + ((c[i] >> 3) & cmask)), 1u << (c[i] & 31u));
                }
            }
        }
        // the first stage's values: requested now, they land while the bitmap is cleared
        u32x4 v0;
        {
            const int off = __builtin_amdgcn_readlane(myd_x, 0), cnt = __builtin_amdgcn_readlane(myd_y, 0);
            const int vo = (4 * lane < cnt) ? lane * 16 : (int)(OOB_SOFFSET - (unsigned)off);
            v0 = __builtin_amdgcn_raw_buffer_load_b128(rs_val, vo, off, 0);
        }
        // MATRIX filter (s_plus.h:159-171): the row's excluded columns are marked in the collision bitmap, so all their
        // products gather in the collision set, where the excluded columns are dropped at the scan
        if (p.filter_mode == SP_SEL_MATRIX) {
            const int f0 = __builtin_amdgcn_readfirstlane(p.f_indptr[t]), f1 = __builtin_amdgcn_readfirstlane(p.f_indptr[t + 1]);
            for (int i = f0 + tid; i < f1; i += NT) {
                const unsigned c = (unsigned)p.f_indices[i];
                atomicOr((unsigned *)(cbm + ((c >> 3) & cmask)), 1u << (c & 31u));
            }
        }
        wg_sync<true>();
        PHASE_END(PH_SWEEP1);
        // descriptor two rows ahead
        if (!p.static_sched) q_nn = sh[SH_QA];
        const int dwNN = load_desc(q_nn);
        // the bitmap has done its job: back to zero; its storage now serves sweep 2
#pragma unroll 4
        for (int i = tid; i < (nb_bytes >> 4); i += NT) ((int4 *)rA)[i] = make_int4(0, 0, 0, 0);
        // rank structure of the collision bitmap: pre16[w] = marked columns in the words below w (see sp_sparse_kernel.hpp).
        // CBM_BYTES / 16 = 512 threads hold four words each; the waves' totals are combined by a second scan in every wave.
        {
            static_assert(CBM_BYTES / 16 <= RR_NT, "one trip");
            const int4 w4 = (tid < CBM_BYTES / 16) ? ((const int4 *)cbm)[tid] : make_int4(0, 0, 0, 0);
            const int p0 = __popc((unsigned)w4.x), p1 = p0 + __popc((unsigned)w4.y), p2 = p1 + __popc((unsigned)w4.z);
            const int tot = p2 + __popc((unsigned)w4.w);
            const int incl = wave_incl_scan_dpp(tot);
            if (lane == 63) sh[SH_WSUM + wave] = incl;
            wg_sync<true>();
            const int ws = (lane < NW) ? sh[SH_WSUM + lane] : 0;
            const int ws_incl = wave_incl_scan_dpp(ws);
            const int all = __builtin_amdgcn_readlane(ws_incl, 63);
            const int woff = __builtin_amdgcn_readlane(ws_incl - ws, wave);
            const int ex = woff + incl - tot;
            if (tid < CBM_BYTES / 16) {
                const u64 packed = (u64)(unsigned)(ex & 0xFFFF) | ((u64)(unsigned)((ex + p0) & 0xFFFF) << 16) |
                                   ((u64)(unsigned)((ex + p1) & 0xFFFF) << 32) | ((u64)(unsigned)((ex + p2) & 0xFFFF) << 48);
                ((u64 *)pre16)[tid] = packed;
            }
            if (all > CSN / 2) failed = true;      // more marked columns than direct slots (uniform)
        }
        PHASE_END(PH_SEGMENTS);

        if (!failed) {
            // Stages: as in sp_sparse_kernel.hpp, in units of SLOTS (16 items, one per wave).  MONO: survivors of a sweep go
            // straight into the candidate buffer keyed by the raw dot, products of marked columns to the member pool.
            const int room = cap - min(p.k, cap - 1);
            int j0 = 0;                      // slots [0, j0) are done ...
            bool slot0_pending = false;      // ... except this wave's slot 0 (first-stage fallback)
            int chunk_slots = max(1, (room / ITEM) / NW);
            bool last_stage = false;
            bool force_sel = false;
            WavePool wpm{0, -1};

            // ---- first stage without any selection (see sp_sparse_kernel.hpp): one item per wave, the cutoff is the minimum
            // over the waves of the m-th largest per-lane maximum; counted exactly after one more barrier ----
            {
                float v[4], x[4];
                u64 M[4], S[4];
                unsigned lmax = 0u;
                const unsigned c[4] = {ids[0].x, ids[0].y, ids[0].z, ids[0].w};
                const int cntA = __builtin_amdgcn_readlane(myd_y, 0);
                const float segv = __uint_as_float((unsigned)__builtin_amdgcn_readlane(myd_z, 0));
                v[0] = __uint_as_float(v0.x); v[1] = __uint_as_float(v0.y); v[2] = __uint_as_float(v0.z); v[3] = __uint_as_float(v0.w);
                s2_core(c, v, segv, cutx, x, M, S);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u64 ok = (cntA == ITEM) ? ~0ull : __ballot(4 * lane + j < cntA);
                    M[j] &= ok;
                    S[j] &= ok & ~M[j];
                    if ((S[j] >> lane) & 1ull) lmax = max(lmax, fkey(x[j]));
                }
                // rounds in proportion to the wave's share of the stage's lanes, k + 2*NW ranks in total
                const int mine = (cntA + 3) / 4;
                const int my_rounds = max(1, min(24, ((p.k + 2 * NW) * mine + l16 - 1) / l16));
                unsigned rest = lmax, tw = 0u;
                for (int r = 0; r < my_rounds; ++r) {
                    const unsigned mx = wave_max_u32(rest);
                    if (mx != 0u) tw = mx;
                    rest = (rest >= mx) ? 0u : rest;
                }
                if (lane == 0 && tw != 0u) atomicMin((unsigned *)&sh[SH_SEL], tw);
                wg_sync<true>();
                const unsigned g = (unsigned)sh[SH_SEL];
                u64 G[4];
                int cw = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    G[j] = S[j] & __ballot(fkey(x[j]) >= g);
                    cw += __popcll(G[j]);
                }
                if (lane == 0 && cw) atomicAdd(&sh[SH_NEED], cw);
                wg_sync<true>();
                const int totalA = sh[SH_NEED];
                const bool fits = totalA <= room / 2 && totalA >= p.k;
                const int nfull = max(1, min(NW, (room / 2) / ITEM));      // fallback: the first nfull waves accept everything
                if (fits || wave < nfull) {
                    if (!fits) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) G[j] = S[j];
                    }
                    const int m0 = __popcll(M[0]), m1 = __popcll(M[1]), m2 = __popcll(M[2]), m3 = __popcll(M[3]);
                    if (m0 + m1 + m2 + m3) {
                        if (pool_reserve(wpm, m0 + m1 + m2 + m3, &sh[SH_MCTR], mpcap, &sh[SH_OVF])) {
                            int pos = wpm.pos;
                            lds_push64(M[0], __float_as_uint(x[0]), c[0] + 1u, pos, mpool_off); pos += m0;
                            lds_push64(M[1], __float_as_uint(x[1]), c[1] + 1u, pos, mpool_off); pos += m1;
                            lds_push64(M[2], __float_as_uint(x[2]), c[2] + 1u, pos, mpool_off); pos += m2;
                            lds_push64(M[3], __float_as_uint(x[3]), c[3] + 1u, pos, mpool_off);
                            wpm.pos = pos + m3;
                        }
                    }
                    const int n0 = __popcll(G[0]), n1 = __popcll(G[1]), n2 = __popcll(G[2]), n3 = __popcll(G[3]);
                    if (n0 + n1 + n2 + n3) {
                        int ubase = 0;
                        if (lane == 0) ubase = atomicAdd(&sh[SH_CNT], n0 + n1 + n2 + n3);
                        int pos = __builtin_amdgcn_readfirstlane(ubase);
                        if (pos + n0 + n1 + n2 + n3 <= cap) {      // (the fallback of a k close to cap could run over)
                            if ((G[0] >> lane) & 1ull) U[pos + mbcnt64(G[0])] = ((u64)fkey(x[0]) << 32) | (u64)c[0];
                            pos += n0;
                            if ((G[1] >> lane) & 1ull) U[pos + mbcnt64(G[1])] = ((u64)fkey(x[1]) << 32) | (u64)c[1];
                            pos += n1;
                            if ((G[2] >> lane) & 1ull) U[pos + mbcnt64(G[2])] = ((u64)fkey(x[2]) << 32) | (u64)c[2];
                            pos += n2;
                            if ((G[3] >> lane) & 1ull) U[pos + mbcnt64(G[3])] = ((u64)fkey(x[3]) << 32) | (u64)c[3];
                        } else if (lane == 0) sh[SH_OVF] = 1;
                    }
                } else slot0_pending = true;
                j0 = 1;
                if (fits) {
                    have_thr = true;
                    thr_key = g;
                    cutx = fmaxf(cutx0, funkey(g));
                }
                wg_sync<true>();
                if (sh[SH_OVF]) failed = true;
                {
                    const float left = (float)(cap - min(sh[SH_CNT], cap));
                    const float pos = 4.f * (float)l16 * (fits ? 1.f : (float)nfull * (1.f / NW));      // products offered so far
                    const float ch = fits ? 2.f * pos * left / (float)max(2 * p.k, totalA) : 0.5f * left;
                    chunk_slots = max(1, (int)fminf(ch * (1.f / (ITEM * NW)), 1e6f));
                    force_sel = fits && totalA > 8 * p.k;
                }
                PHASE_END(PH_SWEEP2);
            }

            while (!last_stage && !failed) {
                const int j1 = (force_sel && j0 < n_slots) ? j0 : min(n_slots, j0 + chunk_slots);
                // this wave's slots of the stage: [js, je) (a wave whose slot 0 is still pending takes it along, whatever j1 is)
                const int js = slot0_pending ? 0 : j0;
                const int je = slot0_pending ? max(j1, 1) : j1;
                slot0_pending = false;
                if (je > js) {
                    // ---- sweep 2 over slots [js, je): values only, RR_PF loads in flight.  Fully unrolled over the slots (the
                    // resident column ids are registers: static indices), with every slot's LOAD outside its guard — a slot that
                    // is not part of the stage gets an out-of-range load that fetches nothing — so that the loads in flight at a
                    // slot's body are the same on every path and the waits are exact (vmcnt(RR_PF - 1)); only the bodies are guarded. ----
                    WavePool wps{0, -1};
                    auto vload = [&](int j) __attribute__((always_inline)) -> u32x4 {
                        const int off = __builtin_amdgcn_readlane(myd_x, j), cnt = __builtin_amdgcn_readlane(myd_y, j);
                        const int lim = (j >= js && j < je) ? cnt : 0;      // not in the stage: nothing is fetched
                        const int vo = (4 * lane < lim) ? lane * 16 : (int)(OOB_SOFFSET - (unsigned)off);
                        return __builtin_amdgcn_raw_buffer_load_b128(rs_val, vo, off, 0);
                    };
                    auto slot_body = [&](int j, const u32x4 &cc, const u32x4 &vv) __attribute__((always_inline)) {
                        if (j < js || j >= je) return;
                        const int cnt = __builtin_amdgcn_readlane(myd_y, j);
                        if (cnt == 0) return;      // (an unused slot of this wave: wave-uniform)
                        const float segv = __uint_as_float((unsigned)__builtin_amdgcn_readlane(myd_z, j));
                        const unsigned c[4] = {cc.x, cc.y, cc.z, cc.w};
                        const float v[4] = {__uint_as_float(vv.x), __uint_as_float(vv.y), __uint_as_float(vv.z), __uint_as_float(vv.w)};
                        __builtin_amdgcn_s_setprio(3);
                        float x[4];
                        u64 M[4], S[4];
                        s2_core(c, v, segv, cutx, x, M, S);
                        if (cnt != ITEM) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const u64 ok = __ballot(4 * lane + i < cnt);
                                M[i] &= ok;
                                S[i] &= ok;
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) S[i] &= ~M[i];
                        if ((M[0] | M[1]) | (M[2] | M[3])) {
                            const int n0 = __popcll(M[0]), n1 = __popcll(M[1]), n2 = __popcll(M[2]), n3 = __popcll(M[3]);
                            if (pool_reserve(wpm, n0 + n1 + n2 + n3, &sh[SH_MCTR], mpcap, &sh[SH_OVF])) {
                                int pos = wpm.pos;
                                lds_push64(M[0], __float_as_uint(x[0]), c[0] + 1u, pos, mpool_off); pos += n0;
                                lds_push64(M[1], __float_as_uint(x[1]), c[1] + 1u, pos, mpool_off); pos += n1;
                                lds_push64(M[2], __float_as_uint(x[2]), c[2] + 1u, pos, mpool_off); pos += n2;
                                lds_push64(M[3], __float_as_uint(x[3]), c[3] + 1u, pos, mpool_off);
                                wpm.pos = pos + n3;
                            }
                        }
                        if ((S[0] | S[1]) | (S[2] | S[3])) {
                            const int n0 = __popcll(S[0]), n1 = __popcll(S[1]), n2 = __popcll(S[2]), n3 = __popcll(S[3]);
                            if (pool_reserve<16>(wps, n0 + n1 + n2 + n3, &sh[SH_CNT], cap, &sh[SH_OVF])) {
                                int pos = wps.pos;
                                lds_push64(S[0], c[0], fkey(x[0]), pos, u_off); pos += n0;
                                lds_push64(S[1], c[1], fkey(x[1]), pos, u_off); pos += n1;
                                lds_push64(S[2], c[2], fkey(x[2]), pos, u_off); pos += n2;
                                lds_push64(S[3], c[3], fkey(x[3]), pos, u_off);
                                wps.pos = pos + n3;
                            }
                        }
                        __builtin_amdgcn_s_setprio(0);
                    };
                    u32x4 vb[RR_PF];
#pragma unroll
                    for (int i = 0; i < RR_PF; ++i) vb[i] = vload(i);
#pragma unroll
                    for (int j = 0; j < RR_NSLOT; ++j) {
                        slot_body(j, ids[j], vb[j % RR_PF]);
                        if (j + RR_PF < RR_NSLOT) vb[j % RR_PF] = vload(j + RR_PF);
                    }
                }
                j0 = j1;
                last_stage = (j0 >= n_slots);
                if (last_stage) break;      // (the finish follows the loop)
                wg_sync<true>();
                if (sh[SH_OVF]) { failed = true; break; }     // a pool overflowed
                PHASE_END(PH_SWEEP2);
                // between stages: a selection when U is filling up — it raises the running k-th value, the cutoff of the next stage
                {
                    const int n_eff = min(sh[SH_CNT], cap);
                    const bool want_sel = n_eff > p.k && (!have_thr || force_sel || 2 * n_eff > cap + p.k);
                    force_sel = false;
                    if (want_sel) {
                        const long long thr_new = select_fast<NT, true, SEL_E, true>(U, hist4, sh, p.k, false, 0u, tid);
                        if (thr_new >= 0) {
                            have_thr = true;
                            thr_key = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)thr_new);
                            cutx = fmaxf(cutx0, funkey(thr_key));
                        }
                        PHASE_END(PH_SELECT);
                    }
                }
                // next chunk (see sp_sparse_kernel.hpp): k * m / pos of the next m products pass the k-th best of pos products
                const float pos = (float)macs32 * fminf(1.f, (float)(j0 * NW) / (float)max(1, n_items));
                const float left = (float)max(64, cap - min(sh[SH_CNT], cap));
                const float cnt_u = (float)max(2 * p.k, min(sh[SH_CNT], cap));
                const float ch = have_thr ? fmaxf((float)ITEM, 2.f * pos * left / cnt_u) : (float)room;
                chunk_slots = max(1, (int)fminf(ch * (1.f / (ITEM * NW)), 1e6f));
            }
        }
        // queue position four rows ahead: claimed before the id loads go out (its wait then never includes them), parked in LDS at
        // the end of the row
        int claimed = 0;
        if (!p.static_sched && tid == 0) claimed = (int)atomicAdd(&p.rq[0], 1u);
        // The column ids are dead: the next row's are requested here — the one place, in straight-line code — and land during
        // the dense phases below.
        issue_ids(mydN_x, mydN_y);

        if (!failed) {
            wg_sync<true>();
            if (sh[SH_OVF]) failed = true;     // a pool overflowed
            PHASE_END(PH_SWEEP2);
        }
        if (!failed) {
            const int mext = min(sh[SH_MCTR], mpcap);
            // ---- products of marked columns: find-or-insert in the collision set (see sp_sparse_kernel.hpp) ----
            auto next_slot = [&](unsigned h, unsigned key) __attribute__((always_inline)) -> unsigned {
                const unsigned half = (unsigned)(CSN / 2);
                return (h < half) ? half + (hash_bits((int)key, 2654435761u, cs_shift + 1)) : half + ((h + 1u) & (half - 1u));
            };
            for (int base = 0; base < mext; base += 2 * NT) {
                const int i0m = base + tid, i1m = base + NT + tid;
                const u64 e0 = (i0m < mext) ? mpool[i0m] : 0ull;
                const u64 e1 = (i1m < mext) ? mpool[i1m] : 0ull;
                if (e0 != 0ull) mpool[i0m] = 0ull;
                if (e1 != 0ull) mpool[i1m] = 0ull;
                const unsigned k0 = (unsigned)(e0 >> 32), k1 = (unsigned)(e1 >> 32);
                const float x0 = __uint_as_float((unsigned)e0), x1 = __uint_as_float((unsigned)e1);
                const unsigned c0m = k0 - 1u, c1m = k1 - 1u;
                const unsigned wi0 = (c0m >> 5) & (unsigned)(CBM_BYTES / 4 - 1), wi1 = (c1m >> 5) & (unsigned)(CBM_BYTES / 4 - 1);
                const unsigned bw0 = ((const unsigned *)cbm)[wi0], bw1 = ((const unsigned *)cbm)[wi1];
                unsigned h0 = (unsigned)pre16[wi0] + (unsigned)__popc(bw0 & ((1u << (c0m & 31u)) - 1u));
                unsigned h1 = (unsigned)pre16[wi1] + (unsigned)__popc(bw1 & ((1u << (c1m & 31u)) - 1u));
                bool a0 = (e0 != 0ull), a1 = (e1 != 0ull);
                u64 cur0 = 0ull, cur1 = 0ull, want0 = e0, want1 = e1;
                int rounds = 0;
                while (__ballot(a0 | a1)) {
                    u64 r0 = 0ull, r1 = 0ull;
                    if (a0) r0 = atomicCAS(&cs[h0], cur0, want0);
                    if (a1) r1 = atomicCAS(&cs[h1], cur1, want1);
                    if (a0) {
                        if (r0 == cur0) a0 = false;
                        else if ((unsigned)(r0 >> 32) == k0) {
                            if (cur0 == 0ull) {
                                cur0 = r0;
                                want0 = (r0 & 0xFFFFFFFF00000000ull) | (u64)__float_as_uint(__uint_as_float((unsigned)r0) + x0);
                            } else { atomicAdd((float *)&cs[h0], x0); a0 = false; }
                        } else { h0 = next_slot(h0, k0); cur0 = 0ull; want0 = e0; }
                    }
                    if (a1) {
                        if (r1 == cur1) a1 = false;
                        else if ((unsigned)(r1 >> 32) == k1) {
                            if (cur1 == 0ull) {
                                cur1 = r1;
                                want1 = (r1 & 0xFFFFFFFF00000000ull) | (u64)__float_as_uint(__uint_as_float((unsigned)r1) + x1);
                            } else { atomicAdd((float *)&cs[h1], x1); a1 = false; }
                        } else { h1 = next_slot(h1, k1); cur1 = 0ull; want1 = e1; }
                    }
                    if (++rounds > 4 * CS_MAXPROBE) { sh[SH_OVF] = 1; break; }
                }
            }
            wg_sync<true>();
            if (sh[SH_OVF]) failed = true;     // collision set full
            PHASE_END(PH_ACCUM);
        }
        if (!failed) {
            // ---- the collision set's slots (complete sums) above the cutoff go straight into U; consumed entries are zeroed
            // and their collision-bitmap bits cleared; a full U triggers a selection and another pass over what is left ----
            const int n_ent = CSN;
            for (;;) {
                for (int base = 0; base < n_ent; base += 4 * NT) {
                    u64 e[4];
                    bool want[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = base + j * NT + tid;
                        e[j] = (i < n_ent) ? cs[i] : 0ull;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) want[j] = (e[j] != 0ull) && !(__uint_as_float((unsigned)e[j]) <= cutx);
                    if (p.filter_mode == SP_SEL_MATRIX) {      // (uniform) excluded columns of this row
                        const int f0 = __builtin_amdgcn_readfirstlane(p.f_indptr[t]), f1 = __builtin_amdgcn_readfirstlane(p.f_indptr[t + 1]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (want[j] && range_has(p.f_indices, f0, f1, (int)((unsigned)(e[j] >> 32) - 1u))) want[j] = false;
                    }
                    const u64 m0 = __ballot(want[0]), m1 = __ballot(want[1]), m2 = __ballot(want[2]), m3 = __ballot(want[3]);
                    const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
                    int wbase = 0;
                    if ((m0 | m1) | (m2 | m3)) {
                        if (lane == 0) wbase = atomicAdd(&sh[SH_CNT], n0 + n1 + n2 + n3);
                        wbase = __builtin_amdgcn_readfirstlane(wbase);
                    }
                    const int off[4] = {0, n0, n0 + n1, n0 + n1 + n2};
                    const u64 mm[4] = {m0, m1, m2, m3};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (e[j] != 0ull) {
                            const unsigned col = (unsigned)(e[j] >> 32) - 1u;
                            bool finished = true;
                            if (want[j]) {
                                const int pos = wbase + off[j] + mbcnt64(mm[j]);
                                if (pos < cap) U[pos] = ((u64)fkey(__uint_as_float((unsigned)e[j])) << 32) | (u64)col;
                                else { sh[SH_RETRY] = 1; finished = false; }
                            }
                            if (finished) {
                                cs[base + j * NT + tid] = 0ull;
                                atomicAnd((unsigned *)(cbm + ((col >> 3) & cmask)), ~(1u << (col & 31u)));
                            }
                        }
                    }
                }
                wg_sync<true>();
                const int retry = sh[SH_RETRY];
                const int n_now = sh[SH_CNT];
                wg_sync<true>();
                if (tid == 0) {
                    sh[SH_MCTR] = 0;
                    if (retry) { sh[SH_RETRY] = 0; if (n_now > cap) sh[SH_CNT] = cap; }
                }
                wg_sync<true>();
                PHASE_END(PH_DRAIN);
                const int n_eff = min(n_now, cap);
                if (retry || n_eff > p.k) {
                    const long long thr_new = select_fast<NT, true, SEL_E, true>(U, hist4, sh, p.k, !retry, have_thr ? thr_key : 0u, tid);
                    if (thr_new >= 0) {
                        have_thr = true;
                        thr_key = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)thr_new);
                        cutx = fmaxf(cutx0, funkey(thr_key));
                    }
                    PHASE_END(PH_SELECT);
                }
                if (!retry) break;
            }
        }

        if (!failed) {
            // ================= write-out: epilogue on the winners (s_plus.h:129-156 with the column term folded in: val =
            // xy / den, or the raw dot), exact threshold test, compaction to the front of the slot =================
            wg_sync<true>();
            const int n_sel = min(sh[SH_CNT], p.k);
            const long long o = (long long)slot_i * (long long)p.k;
            if (tid == 0) sh[SH_SEL] = 0;
            wg_sync<true>();
            for (int base = 0; base < n_sel; base += NT) {
                const int j = base + tid;
                const u64 it = (j < n_sel) ? U[j] : 0ull;
                const float xv = funkey((unsigned)(it >> 32));
                float val = xv;
                if (any_norm) val = (den != 0.f) ? xv / den : 0.f;
                const bool keep = (it != 0ull) && (val >= p.threshold);
                const u64 m = __ballot(keep);
                if (m) {
                    int wbase = 0;
                    if (lane == 0) wbase = atomicAdd(&sh[SH_SEL], __popcll(m));
                    wbase = __builtin_amdgcn_readfirstlane(wbase);
                    if (keep) {
                        const long long q = o + wbase + mbcnt64(m);
                        if (p.rows) p.rows[q] = t;
                        p.cols[q] = (int)(unsigned)(it & 0xFFFFFFFFull);
                        p.values[q] = val;
                    }
                }
            }
            wg_sync<true>();
            const int n_out = sh[SH_SEL];
            for (int j = n_out + tid; j < p.k; j += NT) {
                if (p.rows) p.rows[o + j] = 0;
                p.cols[o + j] = 0;
                p.values[o + j] = 0.f;
            }
            if (tid == 0 && p.counts) p.counts[slot_i] = n_out;
            if (p.filter_mode == SP_SEL_MATRIX) {      // marks of excluded columns that no product reached
                const int f0 = __builtin_amdgcn_readfirstlane(p.f_indptr[t]), f1 = __builtin_amdgcn_readfirstlane(p.f_indptr[t + 1]);
                for (int i = f0 + tid; i < f1; i += NT) {
                    const unsigned c = (unsigned)p.f_indices[i];
                    atomicAnd((unsigned *)(cbm + ((c >> 3) & cmask)), ~(1u << (c & 31u)));
                }
            }
            // U's storage is part of the next row's bitmap: every selection zeroes what lies behind the entries it keeps,
            // so only the first k entries can be non-zero here
            wg_sync<true>();
            const int dirty = min(cap, p.k + 2);
            for (int i = tid; i < (dirty + 1) / 2; i += NT) ((int4 *)U)[i] = make_int4(0, 0, 0, 0);
            if (timing) { ph[CT_ROWS_SPARSE] += 1; ph[PH_CSDRAIN] += 1; }      // (slot 8: rows finished by THIS kernel)
        } else {
            // a pool or the collision set overflowed: hand the row to the generic kernel's queue, LDS state back to clean
            wg_sync<true>();
            if (wave == 0) {
                unsigned g = 0;
                if (tid == 0) {
                    g = atomicAdd(&p.qcount[1], 1u);
                    sh[SH_OVF] = 0; sh[SH_RETRY] = 0; sh[SH_CNT2] = 0; sh[SH_EQ] = 0;
                }
                g = (unsigned)__builtin_amdgcn_readfirstlane((int)g);
                if (lane < 8) ((int *)p.desc_g)[8 * (size_t)g + lane] = dwC;
            }
            for (int i = tid; i < (CBM_BYTES + PRE_BYTES + A_bytes) / 16; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
            for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
            if (timing) ph[CT_ROWS_FALLBACK] += 1;
        }
        if (!p.static_sched && tid == 0) sh[SH_NITEMS] = claimed;
        // rotate the row pipeline
        q_c = q_n; q_n = q_nn;
        if (p.static_sched) q_nn += (int)gridDim.x;
        dwC = dwN; dwN = dwNN;
        myd_x = mydN_x; myd_y = mydN_y; myd_z = mydN_z;
        wg_sync<true>();
        PHASE_END(PH_OUTPUT);
    }
    if (timing) {
#pragma unroll
        for (int i = 0; i < PH_N; ++i) atomicAdd(&p.phase_cycles[i], ph[i]);
    }
#undef PHASE_END
}

}  // namespace
