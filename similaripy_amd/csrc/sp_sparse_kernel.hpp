// sp_sparse_kernel.hpp — rows in which few products share an output column: the KNN / recommender shape the
// headline benchmark has (BASELINE configs[1]: 41k products per row scattered over 1M columns, ~2 % collide).
//
// Replaces the per-thread dense `sums[]` array of s_plus.h:71-127 by a column BITMAP in LDS and two sweeps over
// the row's products:
//   sweep 1 (column ids only): one bit per column (exact while n_cols <= bitmap bits, else columns alias modulo
//     the bitmap size); a product that finds its bit set has its column appended to a duplicate pool;
//   the bitmap is cleared and the duplicate columns become a small collision set + a collision bitmap;
//   sweep 2 (ids + values): ONE bit test per product — products of collision-set columns are appended to a pool
//     and accumulated densely afterwards, every other product is provably the only one of its column and is
//     appended only if its raw dot can still beat the running k-th value;
//   the pool is consumed by dense phases: column terms, epilogue (s_plus.h:129-156), threshold, top-k buffer,
//     selection (replaces the heap of s_plus.h:39-64).
// Work is handed out in ITEMS of <= 256 consecutive elements of one m2 row: row base, count and m1 value are
// scalars, one 16-byte buffer load per lane fetches a whole item (fully coalesced 1 KiB per wave instruction),
// and the loads of the next item are in flight while the current one is processed.
// Rows whose pools overflow are handed to the generic kernel through its queue (never to a CPU path).
#pragma once
#include "sp_common.hpp"

namespace {

template <int NT, bool U_LDS>
__global__ __launch_bounds__(NT) void sp_knn_sparse_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A_bytes = p.T * 8;

    // ---- LDS carve-up (single dynamic array) ----
    // region A [0, T*8)   sweep 1: column bitmap (nb bits);
    //                     sweep 2: [0,A/4) collision bitmap, [A/4,A/2) collision set, [A/2,A) survivor / member pool
    // items[ITEM_CAP]     {m2 byte offset, count, m1 value bits, flat start};  hist4[4][256] radix histograms
    // sh[32], ph[16]      scalars, phase timers;   U[cap] candidate buffer (sweep 1 borrows it for the duplicate pool)
    int4 *items = (int4 *)(smem + A_bytes);
    int *hist4 = (int *)(items + ITEM_CAP);
    int *sh = hist4 + 1024;
    u64 *ph = (u64 *)(sh + 32);
    u64 *U = U_LDS ? (u64 *)(ph + 16) : (p.gU + (size_t)blockIdx.x * (size_t)p.cap);

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

    // region A all zero, histograms zero
    for (int i = tid; i < A_bytes / 16; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
    for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
    if (tid < 32) sh[tid] = 0;
    if (tid < 16) ph[tid] = 0;
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    float ymin_tv = 0.f, ymin_cos = 0.f, ymin_dep = 0.f;
    if (p.bound_ok) {
        if (p.fold) { ymin_cos = 1.f; ymin_dep = 1.f; }     // folded column term: exactly 1 for every column
        else { ymin_tv = p.ymin[0]; ymin_cos = p.ymin[1]; ymin_dep = p.ymin[2]; }
    }

    const bool timing = (p.phase_cycles != nullptr) && tid == 0;
    u64 tmark = timing ? (u64)clock64() : 0;
#define PHASE_END(which) do { if (timing) { const u64 _n = (u64)clock64(); ph[which] += _n - tmark; tmark = _n; } } while (0)

    // ---- row pipeline ----
    // The chain  queue -> descriptor {slot, row, m1 start, m1 length | MACs, X terms} -> m1 entries -> m2 row bounds
    // is four dependent global loads (~1 us each under load).  It is software-pipelined across rows: while row r is
    // processed, the queue slot of row r+3 is claimed, the descriptor of row r+2 is loaded, the m1 entries of
    // row r+1 are loaded (top of the row) and its m2 row bounds fetched (middle of the row).
    const int n_rows = (int)p.qcount[0];
    const int4 *desc = p.desc;
    auto load_desc = [&](int q, int4 &d0, int4 &d1) {
        d0 = make_int4(-1, 0, 0, 0);
        d1 = make_int4(0, 0, 0, 0);
        if (q < n_rows) { d0 = desc[2 * (size_t)q]; d1 = desc[2 * (size_t)q + 1]; }
    };
    int q_nn = 0;      // queue index two rows ahead (static schedule: computed; dynamic: through LDS)
    int pend_q = 0;    // (tid 0) claimed queue index three rows ahead
    int4 dC, dN, wC, wN;   // descriptors (both halves) of the current and the next row
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
    // m1 entry / m2 row bounds of segment `tid` of the current row (rows of this kernel have <= SORT_MAX <= NT entries)
    int my_r0 = 0, my_len = 0;
    float my_v = 0.f;
    if (dC.x >= 0 && tid < dC.w) {
        const int u = p.m1_indices[dC.z + tid];
        my_v = p.m1_data[dC.z + tid];
        my_r0 = p.m2_indptr[u];
        my_len = p.m2_indptr[u + 1] - my_r0;
    }

    for (;;) {
        // row-constant values are wave-uniform: v_readfirstlane moves them to scalar registers
        const int slot_i = __builtin_amdgcn_readfirstlane(dC.x);
        if (slot_i < 0) break;
        const int t = __builtin_amdgcn_readfirstlane(dC.y);
        const int n1 = __builtin_amdgcn_readfirstlane(dC.w);
        const unsigned macs32 = (unsigned)__builtin_amdgcn_readfirstlane(wC.x);

        // prefetch: queue slot three rows ahead, m1 entries of the next row
        if (!p.static_sched && tid == 0) {
            sh[SH_QA] = pend_q;
            pend_q = (int)atomicAdd(&p.queue[0], 1u);
        }
        int nx_u = 0;
        float nx_v = 0.f;
        if (dN.x >= 0 && tid < dN.w) {
            nx_u = p.m1_indices[dN.z + tid];
            nx_v = p.m1_data[dN.z + tid];
        }
        int nx_r0 = 0, nx_len = 0;

        // the duplicate pool borrows U's storage (empty until sweep 2); holes must read zero
        for (int i = tid; i < p.cap / 2; i += NT) ((int4 *)U)[i] = make_int4(0, 0, 0, 0);
        if (tid == 0) { sh[SH_DCTR] = 0; sh[SH_PCTR] = 0; sh[SH_NITEMS] = 0; sh[SH_CNT] = 0; }
        // Segments are visited in descending |m1 value| order: each segment scales its m2 row by its own m1 value,
        // so the heavy segments first make the running k-th value rise early and the survivor rate fall
        // monotonically.  First item and flat start of every segment come from one all-pairs pass spread over the
        // whole workgroup (n1 <= 256): thread (seg, part) adds up the segments that precede `seg`.
        int *keyS = (int *)items, *lenS = keyS + SORT_MAX, *ibS = lenS + SORT_MAX, *fsS = ibS + SORT_MAX;
        if (tid < SORT_MAX) { ibS[tid] = 0; fsS[tid] = 0; }
        if (tid < n1) { keyS[tid] = (int)(__float_as_uint(my_v) & 0x7FFFFFFFu); lenS[tid] = my_len; }
        __syncthreads();
        int4 dNN, wNN;
        if (!p.static_sched) q_nn = sh[SH_QA];
        load_desc(q_nn, dNN, wNN);
        if (p.static_sched) q_nn += (int)gridDim.x;
        {
            const int lg = (n1 <= 64) ? 6 : (n1 <= 128) ? 7 : 8;       // segments padded to a power of two >= 64
            const int seg = tid & ((1 << lg) - 1), part = tid >> lg, parts = NT >> lg;
            if (seg < n1) {
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
        int my_ib = 0, my_fs = 0;
        if (tid < n1) { my_ib = ibS[tid]; my_fs = fsS[tid]; }
        const int n_items = sh[SH_NITEMS];
        __syncthreads();                    // scratch read before the items overwrite it
        bool failed = (n_items >= ITEM_CAP);
        PHASE_END(PH_SETUP);

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
        epi.bA = p.l1 * (p.t1 * epi.xtv + p.t2 * ymin_tv) + p.l2 * epi.xcos * ymin_cos + p.l3 * epi.xdep * ymin_dep + p.stab;
        epi.bB = p.l1 * (1.f - p.t1 - p.t2);
        rc.set_cut(p.threshold);
        rc.f0 = rc.f1 = rc.g0 = rc.g1 = 0;
        if (p.filter_mode == SP_SEL_MATRIX) { rc.f0 = __builtin_amdgcn_readfirstlane(p.f_indptr[t]); rc.f1 = __builtin_amdgcn_readfirstlane(p.f_indptr[t + 1]); }
        if (p.target_mode == SP_SEL_MATRIX) { rc.g0 = __builtin_amdgcn_readfirstlane(p.t_indptr[t]); rc.g1 = __builtin_amdgcn_readfirstlane(p.t_indptr[t + 1]); }

        if (!failed) {
            // sentinel item behind the last one: a prefetch past the end loads nothing (every lane out of range)
            if (tid == NT - 1) items[n_items] = make_int4((int)OOB_SOFFSET, 0, 0, (int)macs32);
            if (tid < n1) {
                int q = 0;
                for (int o = 0; o < my_len; o += ITEM, ++q)
                    items[my_ib + q] = make_int4((my_r0 + o) * 4, min(ITEM, my_len - o), (int)__float_as_uint(my_v), my_fs + o);
            }
            __syncthreads();
            PHASE_END(PH_SEGMENTS);

            // ---- sweep 1: column ids only ----
            {
                WavePool wp{0, -1};
                // One 16-byte buffer load per lane fetches a whole item (lane l: elements 4l..4l+3); the range check
                // of the buffer resource is per dword (scripts/buffer_oob_probe.hip), so an item at the very end of
                // the array is safe, and a prefetch past the last item reads the sentinel: an all-out-of-range load
                // (no memory traffic) instead of a branch, so the loads in flight are countable (s_waitcnt vmcnt(N)).
                auto ld = [&](int it, unsigned (&c)[4], int &cnt) __attribute__((always_inline)) {
                    const int4 d = items[min(it, n_items)];
                    const int off = __builtin_amdgcn_readfirstlane(d.x);
                    cnt = __builtin_amdgcn_readfirstlane(d.y);
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, lane * 16, off, 0);
                    c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
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
                            bit[j] = (4 * lane + j < cnt) ? (1u << (c[j] & 31u)) : 0u;     // padding ORs nothing
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
                ld(it, cA, nA);
                while (it < n_items) {
                    ld(it + NW, cB, nB);
                    __builtin_amdgcn_sched_barrier(0);     // the prefetch is issued before the current item is waited for
                    body(cA, nA);
                    if (it + NW >= n_items) break;
                    ld(it + 2 * NW, cA, nA);
                    __builtin_amdgcn_sched_barrier(0);
                    body(cB, nB);
                    it += 2 * NW;
                }
            }
            __syncthreads();
            const int ovf1 = sh[SH_OVF];
            const int dext = min(sh[SH_DCTR], dcap);
            PHASE_END(PH_SWEEP1);
            // next row's m2 row bounds (its m1 entries were requested at the top of this row)
            if (dN.x >= 0 && tid < dN.w) {
                nx_r0 = p.m2_indptr[nx_u];
                nx_len = p.m2_indptr[nx_u + 1] - nx_r0;
            }
            // the bitmap has done its job: back to zero (16-byte stores), then the collision structures go there
            for (int i = tid; i < (nb_bytes >> 4); i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
            __syncthreads();
            failed = (ovf1 != 0);
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
            PHASE_END(PH_ACCUM);     // (bitmap clear + collision-set build)
        } else {
            if (dN.x >= 0 && tid < dN.w) {
                nx_r0 = p.m2_indptr[nx_u];
                nx_len = p.m2_indptr[nx_u + 1] - nx_r0;
            }
        }

        if (!failed) {
            // Stages: sweep 2 over a chunk of items, or (last stage) the collision set itself turned into pool
            // entries; then ONE dense consumer: member products accumulate in the collision set, single products
            // are judged (column terms, epilogue, threshold) and appended to U; a full U triggers a selection and
            // another pass over what is left of the pool.
            // The products are offered in growing chunks with a selection after each: the first chunk is small
            // enough that accepting everything cannot overflow U; once the k-th best of n products is known, about
            // k*m/n of the next m would survive in an exchangeable stream — far fewer here, because segments come
            // in descending weight — so the next chunk may be 4*n*(cap-k)/k long.
            const int room = p.cap - min(p.k, p.cap - 1);
            int i0 = 0;
            long long chunk = room;
            for (;;) {
                const bool last_stage = (i0 >= n_items);     // uniform
                int ext = 0;
                if (!last_stage) {
                    const int i1 = (int)min((long long)n_items, (long long)i0 + max(1ll, chunk / ITEM));
                    // ---- sweep 2 over items [i0, i1) ----
                    WavePool wp{0, -1};
                    auto ld = [&](int it, unsigned (&c)[4], float (&v)[4], int &cnt, float &segv) __attribute__((always_inline)) {
                        const int4 d = items[(it < i1) ? it : n_items];     // beyond this chunk: the sentinel item
                        const int off = __builtin_amdgcn_readfirstlane(d.x);
                        cnt = __builtin_amdgcn_readfirstlane(d.y);
                        segv = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(d.z));
                        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, lane * 16, off, 0);
                        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs_val, lane * 16, off, 0);
                        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
                        v[0] = __uint_as_float(b.x); v[1] = __uint_as_float(b.y); v[2] = __uint_as_float(b.z); v[3] = __uint_as_float(b.w);
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
                            const bool ok = full || (4 * lane + j < cnt);
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
                    ld(it, cA, vA, nA, sA);
                    while (it < i1) {
                        ld(it + NW, cB, vB, nB, sB);
                        __builtin_amdgcn_sched_barrier(0);     // the prefetch is issued before the current item is waited for
                        body(cA, vA, nA, sA);
                        if (it + NW >= i1) break;
                        ld(it + 2 * NW, cA, vA, nA, sA);
                        __builtin_amdgcn_sched_barrier(0);
                        body(cB, vB, nB, sB);
                        it += 2 * NW;
                    }
                    i0 = i1;
                    __syncthreads();
                    ext = min(sh[SH_PCTR], pcap);
                    if (sh[SH_OVF]) { failed = true; break; }     // pool overflowed: dropped products cannot be re-offered
                    PHASE_END(PH_SWEEP2);
                } else {
                    // ---- last stage: the collision set (complete sums now) becomes pool entries; set and bitmap bits cleared ----
                    for (int idx = tid; idx < CSN; idx += NT) {
                        const u64 s = cs[idx];
                        u64 e = 0ull;
                        if (s != 0ull) {
                            const unsigned c = ~(unsigned)(s >> 32);
                            e = ((u64)(c + 1u) << 32) | (s & 0xFFFFFFFFull);
                            cs[idx] = 0ull;
                            atomicAnd((unsigned *)(cbm + ((c >> 3) & cmask)), ~(1u << (c & 31u)));
                        }
                        pool[idx] = e;
                    }
                    ext = CSN;
                    __syncthreads();
                    PHASE_END(PH_CSDRAIN);
                }

                // ---- dense consumer (leaves the pool all zero) ----
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
                                        if ((unsigned)(s >> 32) == nc) {
                                            // optimistic 64-bit compare-and-swap of {key : sum + x} (ds_cmpst_rtn_b64 retires
                                            // 10x the lanes of ds_add_f32 on gfx950); a lost race re-reads
                                            u64 cur = s;
                                            for (;;) {
                                                const u64 want = (cur & 0xFFFFFFFF00000000ull) | (u64)__float_as_uint(__uint_as_float((unsigned)cur) + xy[j]);
                                                const u64 got = atomicCAS(&cs[h], cur, want);
                                                if (got == cur) break;
                                                cur = got;
                                            }
                                            single = false;
                                            break;
                                        }
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
                    const int n_now = sh[SH_CNT];
                    __syncthreads();
                    if (tid == 0) {
                        sh[SH_PCTR] = 0;
                        if (retry) { sh[SH_RETRY] = 0; if (n_now > p.cap) sh[SH_CNT] = p.cap; }   // failed appends over-counted
                    }
                    __syncthreads();         // counter fix-ups visible before the next pushes / the selection
                    PHASE_END(PH_DRAIN);
                    // selection: forced when U overflowed; after the last stage exact (final top-k); between stages
                    // whenever it can raise the running k-th value, except right before the last stage if U has room
                    const bool before_last = (i0 >= n_items) && !last_stage;
                    const bool want_sel = retry || (last_stage ? (n_now > p.k) : (n_now > p.k && (!before_last || 2 * n_now > p.cap + p.k)));
                    if (want_sel) {
                        long long thr_new;
                        if (p.cap <= 2 * NT) thr_new = select_fast<NT>(U, hist4, sh, p.k, last_stage && !retry);
                        else thr_new = compact_topk<NT>(U, hist4, sh, p.k);
                        if (thr_new >= 0) {
                            rc.have_thr = true;
                            rc.thr_key = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)thr_new);
                            rc.set_cut(p.threshold);
                        }
                        PHASE_END(PH_SELECT);
                    }
                    if (!retry) break;  // uniform
                }
                if (last_stage) break;
                const long long pos = (i0 < n_items) ? (long long)items[i0].w : (long long)macs32;
                chunk = rc.have_thr ? max((long long)room, 4ll * pos * (long long)room / (long long)p.k) : (long long)room;
            }
        }

        if (!failed) {
            // ================= write-out =================
            __syncthreads();
            const int n_out = min(sh[SH_CNT], p.k);
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
            if (timing) ph[CT_ROWS_SPARSE] += 1;
        } else {
            // a pool or the collision set overflowed (or the row has too many items): hand the row to the generic
            // kernel's queue and put the LDS state back to clean
            __syncthreads();
            if (tid == 0) {
                const unsigned g = atomicAdd(&p.qcount[1], 1u);
                p.desc_g[2 * (size_t)g] = dC;
                p.desc_g[2 * (size_t)g + 1] = wC;
                sh[SH_OVF] = 0; sh[SH_RETRY] = 0; sh[SH_CNT2] = 0; sh[SH_EQ] = 0;
            }
            for (int i = tid; i < A_bytes / 16; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
            for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
            if (timing) ph[CT_ROWS_FALLBACK] += 1;
        }
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

}  // namespace
