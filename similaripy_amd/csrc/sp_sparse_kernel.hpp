// sp_sparse_kernel.hpp — rows in which few products share an output column: the KNN / recommender shape the
// headline benchmark has (BASELINE configs[1]: 41k products per row scattered over 1M columns, ~2 % collide).
//
// Replaces the per-thread dense `sums[]` array of s_plus.h:71-127 by a column BITMAP in LDS and two sweeps over
// the row's products:
//   sweep 1 (column ids only): one bit per column (exact while n_cols <= bitmap bits, else columns alias modulo
//     the bitmap size); a product that finds its bit set marks its column in a second, small "collision bitmap";
//   the big bitmap is cleared; its storage becomes the collision set, the survivor pool and the top-k buffer;
//   sweep 2 (ids + values): ONE bit test per product — products of marked columns accumulate in the collision
//     set right away (find-or-insert with 64-bit compare-and-swap), every other product is provably the only one
//     of its column and is appended to the pool only if its raw dot can still beat the running k-th value;
//   the pool (and at the end the collision set) is consumed by a dense phase: column terms, epilogue
//     (s_plus.h:129-156), threshold, top-k buffer, selection (replaces the heap of s_plus.h:39-64).
// Work is handed out in ITEMS of <= 256 consecutive elements of one m2 row: row base, count and m1 value are
// scalars, one 16-byte buffer load per lane fetches a whole item (fully coalesced 1 KiB per wave instruction),
// and the loads of the next item are in flight while the current one is processed.
// Rows whose collision set or pool overflow are handed to the generic kernel through its queue (never to a CPU path).
#pragma once
#include "sp_common.hpp"
// The kernel's body is read in eight files: this one (parameters, LDS carve-up, row pipeline and setup, the stage loop's skeleton) and the seven
// phases it includes where they stood — sp_sparse_phase_{sweep1, rank, first_stage, sweep2, accumulate, consume, writeout}.inc.  They are TEXT, not
// functions: same scopes, same locals, and the compiled kernel is byte-identical to the one-file form (round 6: the device code object's .text was
// compared before and after the cut).
#ifndef SP_DUO_FS2
#define SP_DUO_FS2 1
#endif

namespace {

// MONO: the similarity is a monotone function of the raw dot alone — val = xy / den with a per-row den > 0 (cosine-type
// epilogues whose column term is folded into the m2 stream) or val = xy (no normalisation) — and no per-row TARGET
// selector is active (a per-row FILTER is: its columns are pre-marked in the collision bitmap and dropped at the scan).  Then the whole top-k runs on the raw dot: survivors go straight from the sweep into the
// candidate buffer (no survivor pool, no judge phase), the running k-th raw dot IS the cutoff, and the epilogue is
// applied to the k winners at write-out.
// MODE 0: general variant (survivor pool + judge).  MODE 1: MONO.  MODE 2: BND, the BOUNDED variant — a general epilogue (Tversky term,
// additive shrink, several column terms) on the monotone variant's pipeline: the m2 column ids carry a 12-bit code of the column's
// combined term W[c] (BndInfo, sp_common.hpp), so "can this product still matter?" is  x - Kw*W(c) > Q  with two per-row scalars — no
// gather, three instructions more than MONO's compare; what passes goes straight into U as {raw dot, packed id}; in front of every
// selection (and at the row's end) ONE dense pass turns the new entries into {exact value, column} with the column-term gather and the
// epilogue of s_plus.h:129-156 — a few hundred entries per row instead of the general variant's ~6 k judged candidates; cutoffs are
// exact k-th values.  The first stage is MONO's selection-free one on EXACT values: its one trip per wave gathers the column terms of its
// four products per lane and evaluates the epilogue right there (a bound would not do: what the stage does not keep is never offered
// again, so its cutoff must be a value k candidates really reach).  Rows the bound cannot serve (negative row
// terms) go to the generic queue; calls it cannot serve run MODE 0 (sp_knn.hip: both are launched, BndInfo::state picks one on the device).
// DUO (round 6): TWO 512-thread workgroups per CU for rows of the headline's weight (tens of thousands of products over ~10^6 columns).
// One 1024-thread workgroup owns the CU through its 128 KB exact bitmap, and ~25 k of a C2 row's 62 k cycles (accumulate, selection,
// write-out, clear, setup) run with the CU's load stream idle; with two residents the dense phases of one overlap the sweeps of the
// other.  What makes two fit (80 KB each, sp_duo_lds_bytes): the column bitmap has 2^19 bits and ALIASES (columns modulo its size: an
// aliased column is marked like a repeated one and summed per column in the collision set — exact), it OVERLAYS the rank prefix and
// the storage of the collision set / member pool / U, which are all dead during sweep 1; the monotone-type variants need no survivor
// pool, so the collision set takes half of the region (4096 slots: ~1.6 k marked columns per C2 row against 0.8 k with the exact
// bitmap); the member pool (2560 entries against ~4.3 k members per row) is folded into the collision set BETWEEN stages whenever it
// is half full, and a stage is sized so that its expected members fit what is left.
// (SECOND: the same code under a second symbol — the launch of the two-per-CU shape over the queue of heavier rows, in its larger layout (the layout
// itself travels in KParams): profilers then list the two launches of a step apart instead of averaging an 86 ms kernel with an empty 33 us one)
template <int NT, bool U_LDS, int MODE, bool DUO = false, bool SECOND = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(DUO ? 4 : 1, DUO ? 4 : 10))) void sp_knn_sparse_kernel(const KParams p) {
    static_assert(!SECOND || DUO, "only the two-per-CU shape is launched twice");
    constexpr bool MONO = MODE == 1, BND = MODE == 2, MLIKE = MODE != 0;
    static_assert(!DUO || (NT == 512 && U_LDS && MLIKE), "the two-per-CU shape: 512 threads, U in LDS, a monotone-type variant");
    if constexpr (BND) { if (p.bnd->state != 1) return; }                       // (uniform over the grid: written by the per-call passes)
    else if constexpr (!MONO) { if (p.bnd != nullptr && p.bnd->state == 1) return; }   // launched beside the bounded variant: that one runs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    int tid = threadIdx.x;      // (made opaque at every row top, see there)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // region behind the rank prefix: classic T*8 bytes in four quarters; DUO: [collision set 20 KB | member pool 24 KB | U 16 KB]
    const int A_bytes = DUO ? DUO_A_BYTES : p.T * 8;

    // ---- LDS carve-up (single dynamic array) ----
    // cbm[CBM_BYTES]      collision bitmap (columns seen twice in sweep 1), alive through both sweeps; at offset 0 so
    //                     that its reads need no base add;  pre16[]: its per-word popcount prefix (rank of a marked column)
    // region A [.., +A)   sweep 1: column bitmap (nb bits, from the start; DUO: from pre16 on — the prefix is built after the sweep);
    //                     afterwards: [0,A/4) collision set, [A/4,A/2) survivor pool, [A/2,3A/4) member pool, [3A/4,A) candidate buffer U
    //                     (DUO: see above)
    // items[ICAP]         {m2 byte offset, count, m1 value bits, flat start};  hist4[4][256] radix histograms
    // sh[32], ph[16]      scalars, phase timers
    unsigned char *cbm = smem;
    unsigned short *pre16 = (unsigned short *)(smem + CBM_BYTES);      // [CBM_BYTES/4] marked columns below each bitmap word
    unsigned char *rA = smem + CBM_BYTES + PRE_BYTES;
    constexpr int BM_OFF = DUO ? CBM_BYTES : CBM_BYTES + PRE_BYTES;      // LDS byte address of the sweep-1 bitmap
    int4 *items = (int4 *)(rA + A_bytes);
    constexpr int ICAP = DUO ? DUO_ICAP : item_cap(NT);
    int *hist4 = (int *)(items + ICAP);
    int *sh = hist4 + 1024;
    u64 *ph = (u64 *)(sh + 32);
    int *shx = (int *)(ph + PH_N);      // two more scalars (the phase-timer area has 16 slots, PH_N are in use)
    // (DUO: p.T carries the rank-addressed slots of the collision set, p.logT the log2 of its overflow slots, p.cap_s the entries of U —
    // 2048 + 512 | 3072-entry pool | 2048, or, for calls whose rows expect more marks, 3072 + 1024 | 2048-entry pool | 1536: sp_knn.hip make_config)
    const int cs_bytes = DUO ? (p.T + (1 << p.logT)) * 8 : A_bytes / 4;
    const int mp_rel = DUO ? cs_bytes : A_bytes / 2;                           // member pool, relative to region A
    const int u_rel = DUO ? DUO_A_BYTES - p.cap_s * 8 : (A_bytes / 4) * 3;     // candidate buffer U
    u64 *U = U_LDS ? (u64 *)(rA + u_rel) : (p.gU + (size_t)blockIdx.x * (size_t)p.cap_s);
    const int cap = p.cap_s;

    const unsigned amask = (unsigned)((1u << (p.nb_log2 - 3)) - 1u) & ~3u;      // column -> byte of its bitmap word
    const int nb_bytes = 1 << (p.nb_log2 - 3);
    // DUO, more columns than bitmap bits: the bitmap index drops the column's bits 5 .. (s1_core8q); classic: columns modulo the bitmap size
    const unsigned bm_shift = DUO ? 3u + (unsigned)max(0, (32 - __builtin_clz((unsigned)max(p.n_cols, 2) - 1u)) - p.nb_log2) : 3u;
    const unsigned cmask = (unsigned)(CBM_BYTES - 1) & ~3u;                     // column -> byte of its collision-bitmap word
    // a column's mark in the collision bitmap (DUO: two planes, sp_common.hpp)
    auto cbm_mark = [&](unsigned c) __attribute__((always_inline)) {
        if constexpr (DUO) duo_mark(cbm, c);
        else atomicOr((unsigned *)(cbm + ((c >> 3) & cmask)), 1u << (c & 31u));
    };
    auto cbm_unmark = [&](unsigned c) __attribute__((always_inline)) {
        if constexpr (DUO) duo_unmark(cbm, c);
        else atomicAnd((unsigned *)(cbm + ((c >> 3) & cmask)), ~(1u << (c & 31u)));
    };
    // stage length factor (see the chunk rule at the end of a stage: what an exchangeable stream would let into U against what is left of it;
    // far less passes when the segments come in descending weight, the rule).  (Measured for the DUO shape, whose U is half the classic
    // one's: with ONE first-stage trip per wave one row in 30 000 of configs[1] — one in 7 500 of configs[2] — overflowed U and went to the
    // generic queue; a factor of 1 removed them and cost 6 % (one more stage and accumulate pass per row: 85.3 -> 90.3 ms); two first-stage
    // trips per wave — the 4 096 products the classic shape's sixteen waves see — removed them at the factor of 2 for nothing.)
    constexpr float STAGE_FILL = 2.f;
    constexpr int RANK_BYTES = DUO ? DUO_PLANE_BYTES : CBM_BYTES;      // the part of the collision bitmap whose bits have ranks
    u64 *cs = (u64 *)rA;
    const int CSN = cs_bytes / 8;
    // slots [0, CS_DIR) are addressed by rank; a column that finds its rank slot taken by another probes the CS_OVR slots behind them
    const int CS_DIR = DUO ? p.T : CSN / 2, CS_OVR = DUO ? (1 << p.logT) : CSN / 2;
    const int cs_shift = 32 - (DUO ? p.logT : p.logT - 3);                     // 32 - log2(CS_OVR): the hash of a first overflow probe
    u64 *spool = (u64 *)(rA + A_bytes / 4);      // surviving single products of a stage (general variant only)
    const int spcap = A_bytes / 32;
    u64 *mpool = (u64 *)(rA + mp_rel);           // products of marked columns, all stages
    const int mpcap = DUO ? (u_rel - cs_bytes) / 8 : A_bytes / 32;
    // LDS byte addresses for the hand-written cores (they assume the dynamic LDS segment starts at address 0)
    const unsigned mpool_off = (unsigned)(CBM_BYTES + PRE_BYTES + mp_rel);
    const unsigned spool_off = (unsigned)(CBM_BYTES + PRE_BYTES + A_bytes / 4);
    const unsigned u_off = (unsigned)(CBM_BYTES + PRE_BYTES + u_rel);
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(BND ? (void *)p.m2_packed : (void *)p.m2_indices, 0, (int)p.m2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_val = __builtin_amdgcn_make_buffer_rsrc((void *)p.m2_data, 0, (int)p.m2_bytes, 0x00020000);

    // collision bitmap + region A all zero, histograms zero
    for (int i = tid; i < (CBM_BYTES + PRE_BYTES + A_bytes) / 16; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
    for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
    if (MLIKE && !U_LDS) { for (int i = tid; i < cap / 2; i += NT) ((int4 *)U)[i] = make_int4(0, 0, 0, 0); }
    if (tid < 32) sh[tid] = 0;
    if (tid < 16) ph[tid] = 0;
    __syncthreads();

    const bool any_norm = (p.l1 != 0.f || p.l2 != 0.f || p.l3 != 0.f || p.stab != 0.f || p.bayes != 0.f);
    float ymin_tv = 0.f, ymin_cos = 0.f, ymin_dep = 0.f;
    if (p.bound_ok) {
        if (p.fold) { ymin_cos = 1.f; ymin_dep = 1.f; }     // folded column term: exactly 1 for every column
        else { ymin_tv = p.ymin[0]; ymin_cos = p.ymin[1]; ymin_dep = p.ymin[2]; }
    }

#if SP_ABLATION
    if (p.dbg & 128) {      // ablation: workgroups start staggered over ~one row time (are the CUs' phases locked to each other?)
        const u64 until = (u64)clock64() + (u64)((blockIdx.x * 2654435761u) >> 16);      // 0 .. 65535 cycles
        while ((u64)clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
#endif
    // BND: the call's facts, in scalar registers
    float b_rho_tv = 0.f, b_rho_cos = 0.f, b_rho_dep = 0.f, b_ymin_tv = 0.f, b_ymin_cos = 0.f, b_ymin_dep = 0.f;
    if constexpr (BND) {
        // (v_readfirstlane: the loads are vector loads — nothing tells the compiler that the workspace header is not written by this
        // kernel — and their results would sit in vector registers around the whole row loop)
        auto sf = [](float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
        b_rho_tv = sf(p.bnd->rho_tv); b_rho_cos = sf(p.bnd->rho_cos); b_rho_dep = sf(p.bnd->rho_dep);
        b_ymin_tv = sf(p.bnd->ymin_tv); b_ymin_cos = sf(p.bnd->ymin_cos); b_ymin_dep = sf(p.bnd->ymin_dep);
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
    // (the descriptor two rows ahead stays pending through a whole row: ONE register — lane i < 8 holds its dword i — instead
    // of eight registers with the same 32 bytes in every lane; ln: the caller's per-row lane id, so that no per-lane pointer is
    // hoisted out of the row loop)
    auto load_desc_v = [&](int q, int ln, unsigned &dv) {
        dv = (ln == 0) ? 0xFFFFFFFFu : 0u;
        if (q < n_rows && ln < 8) dv = (((const unsigned *)desc) + 8 * (size_t)q)[ln];
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
    // precomputed work items of a row (sp_row_items_kernel): thread i holds record i of the row's block (records 1 ..: the items),
    // loaded a row ahead like the rest of the row pipeline
    constexpr bool REC2 = NT < ITEMS_STRIDE;      // a small workgroup holds two records per thread
    // Row-pipeline loads must stay PENDING until their values are needed a row (or half a row) later.  The compiler ended that in
    // four ways, each a full memory round trip on the row's critical path (found in the ISA, round 3): a load from a uniform
    // address is moved to scalar registers at once (s_waitcnt vmcnt(0) + v_readfirstlane right behind it) — the record counts of
    // a row therefore travel in its DESCRIPTOR (DESC_* below, written by sp_row_items_kernel) instead of a header of their own;
    // an index is sign-extended for its later use as soon as it arrives; a difference of two loaded values is formed where the
    // loads are issued; the atomic optimizer turns one lane's returning atomic into a form that needs the result at once
    // (-amdgpu-atomic-optimizer-strategy=None in _build.py).  A register SPILL does the same: its reload counts in vmcnt.
    // (the records live in native 128-bit vectors so that ONE opaque asm operand can consume them at the row's end — per-component
    // operands split the tuple and the load's result is copied out, i.e. waited for, right behind the load)
    auto load_items = [&](int slot, int n_rec, u32x4 &rec, u32x4 &rec2) {
        rec = u32x4{0u, 0u, 0u, 0u};
        rec2 = u32x4{0u, 0u, 0u, 0u};
        if (n_rec > 0) {
            // (the thread's index is made opaque here: left alone, `items_g + tid` is hoisted out of the row loop as a 64-bit
            // per-thread pointer, which at this kernel's register budget is spilled and reloaded at every row top)
            int t_o = tid;
            asm volatile("" : "+v"(t_o));
            const u32x4 *row = (const u32x4 *)(p.items_g + (size_t)slot * (size_t)p.items_stride);
            if (t_o <= n_rec) rec = row[t_o];          // (record 0: the bounds of the row's MATRIX-filter list, see below)
            if (REC2 && t_o + NT <= n_rec) rec2 = row[t_o + NT];
        }
    };
    constexpr bool PACK_OK = NT == 256;
    // One trip of a wave: piece A = record `a` in lanes [0, sB), piece B (packed trips of the prepass only) in lanes [sB, 64).
    // Per lane: byte offset of its quad in m2, the number of its real elements (<= 0: none), its m1 value.
    auto trip_lane = [&](int offA, int cntA, unsigned svA, int offB, int cntB, unsigned svB, int sB, int &vo, int &d, float &sv) __attribute__((always_inline)) {
        vo = offA + lane * 16;
        d = cntA - 4 * lane;
        sv = __uint_as_float(svA);
        if (lane >= sB) { vo = offB + (lane - sB) * 16; d = cntB - 4 * (lane - sB); sv = __uint_as_float(svB); }
        if (d <= 0) vo = (int)OOB_SOFFSET;          // lanes beyond the pieces fetch nothing
    };
    u32x4 recC, recN, recC2, recN2;
    load_items(dC.x, desc_n_rec(dC.w), recC, recC2);
    // m1 entry / m2 row bounds of segment `tid` of the current row (rows of this kernel have <= SORT_MAX <= NT entries)
    int my_r0 = 0, my_len = 0;
    float my_v = 0.f;
    if (dC.x >= 0 && tid < desc_n1(dC.w)) {
        const int u = p.m1_indices[dC.z + tid];
        my_v = p.m1_data[dC.z + tid];
        my_r0 = p.m2_indptr[u];
        my_len = p.m2_indptr[u + 1] - my_r0;
    }

    // (what was loaded in front of the loop is consumed in front of it: a value still pending at the loop's header makes the compiler
    // wait with vmcnt(0) at its first use in EVERY iteration — in the later ones for the previous row's result stores)
    asm volatile("" : "+v"(recC), "+v"(my_r0), "+v"(my_len), "+v"(my_v));
    if constexpr (REC2) asm volatile("" : "+v"(recC2));
    {
        auto scalar4 = [](int4 v) {
            return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
        };
        dC = scalar4(dC); wC = scalar4(wC); dN = scalar4(dN); wN = scalar4(wN);
    }

    // Barriers of the row loop: everything the waves of a row exchange goes through LDS (U too, when U_LDS), so they wait for
    // LDS traffic only (wg_sync<true>): __syncthreads() is a workgroup-scope fence and drains vmcnt as well, which made every
    // barrier behind a row-pipeline prefetch (next row's m1 entries, descriptors, m2 bounds) and behind the write-out stores
    // wait for those round trips (round 3: measured with the register-resident experiment, profiles/r03_exp_rowreg.txt).
    for (;;) {
        // row-constant values are wave-uniform: v_readfirstlane moves them to scalar registers
        // (the thread index is opaque per row: addresses derived from it are recomputed where they are used instead of being
        // hoisted out of the row loop, held in registers through the sweeps and — at this kernel's budget — spilled)
        asm volatile("" : "+v"(tid));
        const int slot_i = __builtin_amdgcn_readfirstlane(dC.x);
        if (slot_i < 0) break;
        const int t = __builtin_amdgcn_readfirstlane(dC.y);
        const int dw = __builtin_amdgcn_readfirstlane(dC.w);
        const int n1 = desc_n1(dw);
        const unsigned macs32 = (unsigned)__builtin_amdgcn_readfirstlane(wC.x);

        // prefetch: queue slot three rows ahead, m1 entries of the next row
        if (!p.static_sched && tid == 0) {
            sh[SH_QA] = pend_q;
            pend_q = (int)atomicAdd(&p.queue[0], 1u);
        }
        int nx_u = 0;
        float nx_v = 0.f;
        if (dN.x >= 0 && tid < desc_n1(dN.w)) {
            nx_u = p.m1_indices[dN.z + tid];
            nx_v = p.m1_data[dN.z + tid];
        }
        int nx_r0 = 0, nx_r1 = 0;
        load_items(dN.x, (dN.x >= 0) ? desc_n_rec(dN.w) : 0, recN, recN2);
        const int n_pre = desc_n_trips(dw);       // > 0: the row's trips were cut (and packed) by the prepass
        const int n_rec = desc_n_rec(dw);         //      ... into this many records
        // some trips carry a second piece (B records behind the sentinel).  Only the 256-thread shape is launched on packed rows
        // (sp_row_items_kernel's `pack`): the per-lane bookkeeping costs the 1024-thread shape 2.4 % on C2 rows it never packs
        const bool two_piece = PACK_OK && n_pre > 0 && n_rec > n_pre + 1;

        // (MONO: SH_PCTR is the write-out's compaction counter — zero whenever a row reaches its write-out, reset behind every stage's
        // drain — and must not be touched here: the previous row's threads may still be reading it, there is no barrier in between)
        if (tid == 0) { if (!MLIKE) sh[SH_PCTR] = 0; sh[SH_MCTR] = 0; sh[SH_NITEMS] = 0; sh[SH_CNT] = 0; sh[SH_SEL] = -1; sh[SH_NEED] = 0; }
        // Segment order.  The heaviest segments (largest |m1 value|: each segment scales its m2 row by its own m1 value)
        // go first, so that the first stage of sweep 2 sees the large products and the running k-th value — the cutoff
        // of everything after — starts high.
        int my_ib = 0, my_fs = 0;       // first item / flat start of segment `tid`
        int n_items = 0;
        unsigned dNN;         // descriptor two rows ahead, dword i in lane i
        int4 dR = make_int4(-1, 0, 0, 0), wR = make_int4(0, 0, 0, 0);      // ... and moved to scalar registers there, for the rotation
        // a row with prepass records that fit: the records go to LDS right here, in front of the row's ONE setup barrier (the previous row
        // is done with `items`; that barrier also publishes the queue slot and the counters reset above)
        const bool precut_ok = n_pre > 0 && n_pre < ICAP && n_pre <= 63 * NW && n_rec <= ICAP;      // uniform
        if (n_pre > 0) {
            if (precut_ok) {
                // sentinel item behind the last one: a prefetch past the end loads nothing (every lane out of range)
                // (the image holds the sentinel too: the same 16 bytes as this store)
                if (tid == NT - 1) items[n_pre] = make_int4((int)OOB_SOFFSET, 0, 0, (int)macs32);
                if (tid >= 1 && tid <= n_rec) ((u32x4 *)items)[tid - 1] = recC;
                if (REC2 && tid + NT <= n_rec) ((u32x4 *)items)[tid + NT - 1] = recC2;
                if (MLIKE && tid == 0) { shx[0] = (int)recC.x; shx[1] = (int)recC.y; }
            }
            wg_sync<U_LDS>();
            if (!p.static_sched) q_nn = sh[SH_QA];
            load_desc_v(q_nn, tid & 63, dNN);
            if (p.static_sched) q_nn += (int)gridDim.x;
            n_items = n_pre;
        } else if (n1 <= 64) {
            // One wave, one segment per lane, no barrier inside: the (up to) 8 largest |values| are found with 8 wave-max
            // rounds; heavy segments first, the others behind, both in their original order (ballot + mbcnt); item and
            // flat-start prefixes by one trip through LDS into position order and a DPP scan there.
            if (tid < 64) {
                const unsigned key = (tid < n1 && my_len > 0) ? ((__float_as_uint(my_v) & 0x7FFFFFFFu) | 1u) : 0u;   // 0 = no segment
                unsigned rest = key, thr = 0u;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned mx = wave_max_u32(rest);
                    if (mx != 0u) thr = mx;                    // uniform
                    rest = (rest >= mx) ? 0u : rest;
                }
                const bool heavy = key != 0u && key >= thr;
                const u64 H = __ballot(heavy), Lg = __ballot(key != 0u && !heavy);
                const int pos = heavy ? mbcnt64(H) : __popcll(H) + mbcnt64(Lg);
                const int nit = (my_len + ITEM - 1) / ITEM;
                // Scratch (the items are written after it is read back).  The lanes of this wave talk to each other through
                // it without a barrier: LDS executes a wave's accesses in order.  To the compiler that is one thread reading
                // back its own store — for a lane without a segment it folded the read to the 0 just written there and lost
                // the segment another lane had scattered to that position (rows whose m1 entries point at EMPTY m2 rows;
                // found by scripts/fuzz_parity.py).  The wavefront-scope fences emit no instruction; they keep the
                // compiler from forwarding a lane's own store across them.
                int *scr = (int *)items;
                scr[tid] = 0; scr[64 + tid] = 0;
                if (key != 0u) { scr[pos] = nit; scr[64 + pos] = my_len; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int nit_p = scr[tid], len_p = scr[64 + tid];
                const int ib_incl = wave_incl_scan_dpp(nit_p), fs_incl = wave_incl_scan_dpp(len_p);
                scr[128 + tid] = ib_incl - nit_p;
                scr[192 + tid] = fs_incl - len_p;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (key != 0u) { my_ib = scr[128 + pos]; my_fs = scr[192 + pos]; }
                if (tid == 63) sh[SH_NITEMS] = ib_incl;
            }
            wg_sync<U_LDS>();
            if (!p.static_sched) q_nn = sh[SH_QA];
            load_desc_v(q_nn, tid & 63, dNN);
            if (p.static_sched) q_nn += (int)gridDim.x;
            n_items = sh[SH_NITEMS];
            wg_sync<U_LDS>();                    // scratch read before the items overwrite it
        } else {
            // up to SORT_MAX entries: full descending order from one all-pairs pass spread over the whole workgroup:
            // thread (seg, part) adds up the segments that precede `seg`
            int *keyS = (int *)items, *lenS = keyS + SORT_MAX, *ibS = lenS + SORT_MAX, *fsS = ibS + SORT_MAX;
            if (tid < SORT_MAX) { ibS[tid] = 0; fsS[tid] = 0; }
            if (tid < n1) { keyS[tid] = (int)(__float_as_uint(my_v) & 0x7FFFFFFFu); lenS[tid] = my_len; }
            wg_sync<U_LDS>();
            if (!p.static_sched) q_nn = sh[SH_QA];
            load_desc_v(q_nn, tid & 63, dNN);
            if (p.static_sched) q_nn += (int)gridDim.x;
            {
                const int lg = (n1 <= 128) ? 7 : 8;       // segments padded to a power of two
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
            wg_sync<U_LDS>();
            if (tid < n1) { my_ib = ibS[tid]; my_fs = fsS[tid]; }
            n_items = sh[SH_NITEMS];
            wg_sync<U_LDS>();                    // scratch read before the items overwrite it
            // (DUO: the item area is 240 records, the scratch 4 KB: its last 256 bytes are the head of the first radix histogram, which
            // every selection expects to find zero — the next barrier, in front of any selection, publishes the stores)
            if constexpr (DUO) { if (tid < (4 * SORT_MAX * 4 - ICAP * 16) / 4) hist4[tid] = 0; }
        }
        bool failed = (n_items >= ICAP) || (n_items > 63 * NW) || (n_pre > 0 && n_rec > ICAP);      // (a wave keeps its <= 63 item descriptors in one register)
        int n_marks = 0;      // (uniform) marked bits of the collision bitmap, known behind sweep 1
        // why a row is handed to the generic queue (profiling: the upper bytes of the fallback counter — 32..39 too many items / a row the
        // variant cannot serve, 40..47 collision set (marks beyond its rank slots, or full), 48..55 candidate buffer U full, 56..63 member pool full)
        int why = failed ? 0 : -1;
        PHASE_END(PH_SETUP);

        RowCtx rc;
        rc.have_thr = false;
        rc.thr_key = 0;
        float cutx = -__builtin_inff();    // MONO: a single product / a column sum <= cutx cannot enter the top-k
        float cutx0 = cutx;                // MONO: the part of it that comes from the `threshold` parameter
        const float den = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(wC.y));   // MONO: val = xy / den (desc)
        if constexpr (MONO) {
            // x <= cutx0  =>  val(x) < threshold for sure (the exact test is repeated on the winners at write-out)
            if (!any_norm) {
                cutx0 = __uint_as_float(RowCtx::funkey_inv_below(p.threshold));
            } else {
                const float c0 = p.threshold * den;
                cutx0 = c0 - fabsf(c0) * 2e-6f - 1e-37f;
                if (!(c0 == c0)) cutx0 = -__builtin_inff();      // NaN threshold: nothing is pruned here, all dropped at write-out
            }
            cutx = cutx0;
        } else {
            rc.row = t;
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
        }
        // BND: den(x, c) = A + sum_j r_j*Y_j[c] + bB*x  >=  AE + lam*W[c] + bB*x   (BndInfo, sp_common.hpp; bB <= 0, threshold >= 0, a1 = 1 and
        // no Bayesian shrink by dispatch).  A candidate is DEAD when  1.00002*x / (AE + lam*W + bB*x) <= t,  i.e. when  x - Kw*W <= Q  with
        // Kw = t*lam/P, Q = t*AE/P, P = 1.00002 - t*bB > 0 (both shaved so that rounding can only let more through); t = the value a candidate
        // must beat: just below `threshold`, or the running k-th EXACT value.  No t > 0 yet: only negative raw dots are dead (their value
        // is negative, below any threshold >= 0: the row guard below makes the denominator positive for them).
        float b_AE = 0.f, b_lam = 0.f, b_bB = 0.f, b_nKw = 0.f, b_Q = -1e-30f, b_t0 = 0.f;
        auto bnd_cut_for = [&](float tv, float &nKw, float &Q) __attribute__((always_inline)) {      // tv: uniform
            nKw = 0.f; Q = -1e-30f;
            if (tv > 0.f && tv < __builtin_inff()) {
                const float inv = 0.999996f / (1.00002f - tv * b_bB);
                nKw = -(tv * b_lam * inv);
                Q = tv * b_AE * inv;
            }
            nKw = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(nKw)));
            Q = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(Q)));
        };
        auto set_bnd_cut = [&]() __attribute__((always_inline)) {
            float tv = b_t0;
            if (rc.have_thr) tv = fmaxf(tv, funkey(rc.thr_key));
            bnd_cut_for(tv, b_nKw, b_Q);
        };
        int n_keyed = 0;       // BND: U[0, n_keyed) holds {exact value key, column}; entries behind it {raw dot, packed id}
        bool thr_incl = false; // BND: the running k-th value comes from the first stage's statistic — the candidates that REACH it (>=) are
                               // still waiting for the exact pass, which must keep them; after a selection the k best are in U and only
                               // what BEATS the k-th value (>) is wanted
        if constexpr (BND) {
            const Epi &epi = rc.epi;
            const float r_tv = p.l1 * p.t2, r_cos = p.l2 * epi.xcos, r_dep = p.l3 * epi.xdep;
            float lam = __builtin_inff();
            if (b_rho_tv > 0.f) lam = fminf(lam, r_tv / b_rho_tv);
            if (b_rho_cos > 0.f) lam = fminf(lam, r_cos / b_rho_cos);
            if (b_rho_dep > 0.f) lam = fminf(lam, r_dep / b_rho_dep);
            lam *= 0.999998f;
            const float e_tv = fmaxf(0.f, r_tv - lam * b_rho_tv) * b_ymin_tv, e_cos = fmaxf(0.f, r_cos - lam * b_rho_cos) * b_ymin_cos,
                        e_dep = fmaxf(0.f, r_dep - lam * b_rho_dep) * b_ymin_dep;
            const float A = p.l1 * p.t1 * epi.xtv + p.stab;
            b_lam = lam;
            b_AE = (A + ((e_tv + e_cos) + e_dep)) * 0.999998f;
            b_bB = p.l1 * (1.f - p.t1 - p.t2);
            b_t0 = __uint_as_float(RowCtx::funkey_inv_below(p.threshold));
            // rows the bound cannot serve: a negative row term or multiplier, nothing positive in the denominator's bound
            const bool row_ok = p.bound_ok && (r_tv >= 0.f) && (r_cos >= 0.f) && (r_dep >= 0.f) && (A >= 0.f) && (lam >= 0.f) && (lam < __builtin_inff()) &&
                                (b_AE < __builtin_inff()) && (b_AE > 0.f || lam > 0.f) && !(b_bB > 0.f);
            if (!row_ok) { failed = true; why = 0; }
            b_lam = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(b_lam)));
            b_AE = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(b_AE)));
            b_bB = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(b_bB)));
            set_bnd_cut();
        }

        // MONO: the first stage's trips (one or two items per wave, see there) are requested in front of the bitmap's clearing loop and
        // the rank prefix: the round trip (~3.5 k cycles, in which no wave had anything else to do) runs under those ~3.6 k cycles
        // (zero-initialised: undefined on some path, the registers' last contents would be live around the whole row loop)
        constexpr int FS1 = (NT == 256 || (DUO && SP_DUO_FS2)) ? 2 : 1;      // (DUO: eight waves — two trips each show the stage the 4 096 products the classic shape's sixteen waves see)
        constexpr int MAXR1 = (NT == 256) ? 32 : 16;
        u32x4 fsa[FS1], fsb[FS1];
#pragma unroll
        for (int f = 0; f < FS1; ++f) { fsa[f] = u32x4{0u, 0u, 0u, 0u}; fsb[f] = u32x4{0u, 0u, 0u, 0u}; }
        bool fs_early = false;      // uniform
        if (!failed) {
            if (n_pre == 0) {      // (a row with prepass records that is not failed has them in LDS since its setup barrier)
                // sentinel item behind the last one: a prefetch past the end loads nothing (every lane out of range)
                if (tid == NT - 1) items[n_items] = make_int4((int)OOB_SOFFSET, 0, 0, (int)macs32);
                if (tid < n1) {
                    int q = 0;
                    for (int o = 0; o < my_len; o += ITEM, ++q)
                        items[my_ib + q] = make_int4((my_r0 + o) * 4, min(ITEM, my_len - o), (int)__float_as_uint(my_v), my_fs + o);
                }
                wg_sync<U_LDS>();
            }
            PHASE_END(PH_SEGMENTS);
            // MATRIX filter of the monotone variant (s_plus.h:159-171).  The row's excluded columns are (a) marked in the collision
            // bitmap, so all their products gather in the collision set, and (b) given a pseudo-member of value -inf each: the
            // column's sum is then -inf, below any cutoff, and the set's scan drops it like any other low sum — no look-up of the
            // candidate in the filter list (six dependent global loads per candidate before round 3).  A row whose list came with
            // its item records (record 0: first index and length, sp_row_items_kernel) has every thread's column on its way
            // from here on; other rows read the list where they need it.
            int my_fc = -1;
            bool f_regs = false;      // uniform: the list (<= NT columns) is in my_fc
            if constexpr (MLIKE) {
                if (p.filter_mode == SP_SEL_MATRIX && n_pre > 0) {
                    const int f0 = shx[0], fl = shx[1];
                    if (fl <= NT) {
                        f_regs = true;
                        if (tid < fl) my_fc = p.f_indices[f0 + tid];
                    }
                }
            }
            // ==== phase: sweep 1 — column ids only (column bitmap, collision bitmap, the MATRIX filter's marks) ====
#include "sp_sparse_phase_sweep1.inc"
            // ==== phase: bitmap storage back to zero, rank prefix of the collision bitmap ====
#include "sp_sparse_phase_rank.inc"
        }
        // next row's m2 row bounds (its m1 entries were requested at the top of this row).  Requested here, behind the bitmap's
        // clearing loop: in front of it the compiler drained vmcnt at the loop's exit, i.e. wave 0 waited out the round trip
        if (dN.x >= 0 && tid < desc_n1(dN.w)) {
            int u = nx_u;
            asm volatile("" : "+v"(u));      // (the index is needed HERE: no address arithmetic where it was loaded)
            nx_r0 = p.m2_indptr[u];
            nx_r1 = p.m2_indptr[u + 1];
        }

        if (!failed) {
            // Stages.  A stage is a sweep 2 over a chunk of items: products of marked columns go to the member pool,
            // surviving single products to the survivor pool.  After every stage ONE dense consumer judges the
            // survivor pool (column terms, epilogue, threshold) into U; after the last stage the member pool is first
            // accumulated into the collision set (find-or-insert) and the set's slots are judged with the pool; a
            // full U triggers a selection and another pass over what is left.
            // The products are offered in growing chunks with a selection after each: the first chunk is small
            // enough that accepting everything cannot overflow U; once the k-th best of n products is known, about
            // k*m/n of the next m would survive in an exchangeable stream — far fewer here, because segments come
            // in descending weight — so the next chunk may be 4*n*(cap-k)/k long.
            const int room = cap - min(p.k, cap - 1);
            int i0 = 0;
            int chunk_items = max(1, (MLIKE ? room : min(room, spcap - 2 * ITEM)) / ITEM);     // items of the next stage
            bool last_stage = false;
            bool force_sel = false;
            WavePool wpm{0, -1};      // member-pool window: lives across the stages of the row
            int stage_retries = 5;    // (uniform) stages that may be taken back and offered again shorter when a pool overflows, see there
            // MONO: the first trip of the stage BEHIND the selection-free first stage (item pre_i0 + wave) is requested as soon as that
            // stage's own data has arrived, i.e. in front of its statistics rounds and three barriers: the sweep that follows is
            // bound by its bodies, so a trip that is there when it starts moves the whole stage forward by one body
            u32x4 pre_a = u32x4{0u, 0u, 0u, 0u}, pre_b = u32x4{0u, 0u, 0u, 0u};
            int pre_i0 = -1;          // uniform; -1: nothing requested

            // ==== phase: the monotone-type variants' first stage (no selection; cutoff from the waves' per-lane maxima) ====
#include "sp_sparse_phase_first_stage.inc"
            if (i0 >= n_items && !failed && i0 > 0) {
                // (all items went through the first stage: let the loop run its last-stage part with an empty sweep)
            }
            while (!last_stage && !failed) {
                // (force_sel: the first stage's cutoff is loose — far more than k products reached it: tighten it with a
                // selection before sweeping on, i.e. run this round with an empty sweep)
                // DUO: the member pool holds fewer entries than a row has members; it is folded into the collision set between stages
                // (below), and a stage must not bring more members than the pool has room for: a product is a member when its column's bit
                // in both planes of the collision bitmap is marked — by chance ((marks / 32768)^2 of all products) or because its column
                // repeats or shares its bit of the sweep-1 bitmap (up to two products per mark) — hence ITEM * fp + 2 * marks / items members per item; waves reserve whole
                // blocks, so NW blocks of the pool are slack.  (An overflow is not an error: the row goes to the generic queue.)
                int stage_items = chunk_items;
                // Every wave sizes the stage from the pool counters it reads itself (here, and `chunk_items` at the end of the stage before):
                // they must all have read them before any wave's sweep pushes again.  With ONE workgroup per CU the waves leave the
                // barrier in front of those reads together and the first push is a memory round trip away; with two or three, the other
                // workgroups' sweep bodies run at raised priority and can hold a wave of this one back for longer than that — one wave
                // then cut the stage elsewhere than its siblings (round 6, found by the full-size test on the two-per-CU shape: one row in
                // 10^6 with a product counted twice; the three-per-CU 256-thread shape showed it once in a fuzz case of tied values,
                // tests/test_hip_stress.py seed 34 case 48 — not reproducible in six reruns).  Every shape that shares a CU gets the barrier.
                const int m_seen = min(sh[SH_MCTR], mpcap);
                const int u_seen = min(sh[SH_CNT], cap);      // (MLIKE: U and the member pool as the stage finds them — what a retry goes back to)
                const WavePool wpm0 = wpm;
                // (the monotone-type variants of EVERY shape: a stage that overflows is taken back to u_seen / m_seen, which must be the same
                // in every wave — the one-per-CU shape's waves may differ by a memory round trip's worth too, rarely: the race hunt of
                // tests/test_hip_stress.py found one wrong row in 3 000 x 100 on the 1024-thread shape with U in global memory the first
                // time the retry ran without this)
                if constexpr (NT < 1024 || MLIKE) wg_sync<U_LDS>();
                if constexpr (DUO) {
                    const float fm = (float)n_marks * (1.f / (float)(8 * DUO_PLANE_BYTES));      // marked share of a plane's bits
                    const float per_item = 1.25f * ((float)ITEM * fm * fm + 2.f * (float)n_marks * __builtin_amdgcn_rcpf((float)max(1, n_items))) + 2.f;
                    const int room_m = (mpcap - NW * POOL_BLK) - m_seen;
                    stage_items = min(stage_items, max(1, (int)((float)room_m * __builtin_amdgcn_rcpf(per_item))));
                    stage_items = __builtin_amdgcn_readfirstlane(stage_items);
                }
                const int i1 = (force_sel && i0 < n_items) ? i0 : min(n_items, i0 + stage_items);
                {
                    // ==== phase: sweep 2 over the stage's items (ids + values) ====
#include "sp_sparse_phase_sweep2.inc"
                }
                wg_sync<U_LDS>();
                if (sh[SH_OVF]) {     // a pool overflowed
                    // Monotone-type variants: the stage is TAKEN BACK and offered again at a quarter of its length (round 6).  Stage lengths
                    // come from what an exchangeable stream would let through; heavily tied data — binary matrices: every product of a row
                    // is one of a few dozen values — lets far more through, U or the member pool fills, and the row used to go to the
                    // generic queue: 148 k of 200 k rows of a binary matrix with Poisson degrees under Jaccard on the two-per-CU shape
                    // (161 ms against the classic shape's 45: its U is twice the size).  Everything the stage pushed lies behind the
                    // counters' values at its start (U: fresh blocks only; the pool: fresh blocks, and the tail of every wave's own window,
                    // empty then), the cutoff has not moved: zero it, put the counters back, go again.
                    if (MLIKE && stage_retries > 0 && i1 - i0 > 1) {      // (uniform)
                        const int u_now = min(sh[SH_CNT], cap), m_now = min(sh[SH_MCTR], mpcap);
                        wg_sync<U_LDS>();      // every wave has read the counters
                        for (int i = u_seen + tid; i < u_now; i += NT) U[i] = 0ull;
                        for (int i = m_seen + tid; i < m_now; i += NT) mpool[i] = 0ull;
                        {
                            const int wp0 = __builtin_amdgcn_readfirstlane(wpm0.pos), we0 = __builtin_amdgcn_readfirstlane(wpm0.end);
                            if (wp0 + lane < we0) mpool[wp0 + lane] = 0ull;      // (a window is at most POOL_BLK = 64 entries... of what is left of it)
                            for (int i = wp0 + 64 + lane; i < we0; i += 64) mpool[i] = 0ull;
                        }
                        if (tid == 0) { sh[SH_CNT] = u_seen; sh[SH_MCTR] = m_seen; sh[SH_OVF] = 0; }
                        wpm = wpm0;
                        wg_sync<U_LDS>();
                        chunk_items = max(1, (i1 - i0) / 4);
                        --stage_retries;
                        if (timing) ph[CT_PASSES] += 1ull << 53;      // (profiling: stages taken back, bits 53..63 of the counter — modulo 2048; the bounded variant keeps its own counts below)
                        continue;
                    }
                    failed = true; why = (sh[SH_CNT] > cap) ? 2 : 3; break;
                }
                i0 = i1;
                last_stage = (i0 >= n_items);      // (an empty first round with i0 < n_items is never the last)
                const int ext = min(sh[SH_PCTR], spcap);
                const int mext = min(sh[SH_MCTR], mpcap);
                PHASE_END(PH_SWEEP2);
                // (DUO: also between stages, as soon as the pool is half full)
                const bool do_acc = last_stage || (DUO && 2 * mext > mpcap - NW * POOL_BLK);      // uniform
                if (do_acc) {
                    // ==== phase: member pool -> collision set ====
#include "sp_sparse_phase_accumulate.inc"
                }

                // ==== phase: dense consumer, exact pass (bounded variant), selections, next stage's length ====
#include "sp_sparse_phase_consume.inc"
            }
        }

        if (!failed) {
            // ==== phase: write-out ====
#include "sp_sparse_phase_writeout.inc"
        } else {
            // a pool or the collision set overflowed (or the row has too many items): hand the row to the generic
            // kernel's queue and put the LDS state back to clean
            // (this path's copy of the consumption above: on every path in front of the path's own stores)
            asm volatile("" : "+v"(recN), "+v"(nx_v), "+v"(pend_q), "+v"(nx_r0), "+v"(nx_r1), "+v"(dNN));
            if constexpr (REC2) asm volatile("" : "+v"(recN2));
            dR = make_int4(__builtin_amdgcn_readlane((int)dNN, 0), __builtin_amdgcn_readlane((int)dNN, 1),
                           __builtin_amdgcn_readlane((int)dNN, 2), __builtin_amdgcn_readlane((int)dNN, 3));
            wR = make_int4(__builtin_amdgcn_readlane((int)dNN, 4), __builtin_amdgcn_readlane((int)dNN, 5),
                           __builtin_amdgcn_readlane((int)dNN, 6), __builtin_amdgcn_readlane((int)dNN, 7));
            wg_sync<U_LDS>();
            if (tid == 0) {
                const unsigned g = atomicAdd(p.qcount_g, 1u);
                p.desc_g[2 * (size_t)g] = make_int4(dC.x, dC.y, dC.z, n1);      // (without the record counts)
                p.desc_g[2 * (size_t)g + 1] = wC;
                sh[SH_OVF] = 0; sh[SH_RETRY] = 0; sh[SH_CNT2] = 0; sh[SH_EQ] = 0; sh[SH_PCTR] = 0;
            }
            for (int i = tid; i < (CBM_BYTES + PRE_BYTES + A_bytes) / 16; i += NT) ((int4 *)smem)[i] = make_int4(0, 0, 0, 0);
            for (int i = tid; i < 1024; i += NT) hist4[i] = 0;
            if (MLIKE && !U_LDS) { for (int i = tid; i < cap / 2; i += NT) ((int4 *)U)[i] = make_int4(0, 0, 0, 0); }
            if (timing) ph[CT_ROWS_FALLBACK] += 1ull + (1ull << (32 + 8 * max(0, why)));
        }
        // rotate the row pipeline (descriptors are wave-uniform: keep them in scalar registers)
        dC = dN; wC = wN;
        dN = dR; wN = wR;
        my_r0 = nx_r0; my_len = nx_r1 - nx_r0; my_v = nx_v;
        recC = recN; recC2 = recN2;
        // Removed barrier (round 3), the audit VERDICT r3 asked to have written down — every LDS word thread 0 (or any thread) of the NEXT
        // row writes in front of that row's first barrier, and the LAST read of it in this row:
        //   sh[SH_QA]                    written at the next row's top by thread 0;  last read: this row's setup, behind its setup barrier
        //   sh[SH_MCTR], sh[SH_NITEMS]   reset at the top;  last reads: the last stage's `mext` / the setup's `n_items`, each behind a
        //                                barrier that every thread passes before the write-out's closing barrier
        //   sh[SH_CNT]                   reset at the top;  last read: `n_sel` at the write-out's head, IN FRONT of its closing barrier
        //                                (MONO: wg_sync behind the compaction loop; general + LDS U: the "U read before it is cleared"
        //                                barrier) — the variant without such a barrier is the one that keeps the barrier below
        //   sh[SH_SEL], sh[SH_NEED]      reset at the top;  last reads: the first stage / a selection, each closed by its own barrier
        //   sh[SH_PCTR]                  general: reset at the top, last read `ext` behind a stage barrier; MONO: never reset here — it IS
        //                                the write-out's compaction counter, read (n_out) behind the closing barrier and then left at
        //                                the value the next write-out expects only because every stage's drain resets it (see there)
        //   items[] / sort scratch       written by the next row's setup (records, sentinel, scratch);  last reads: this row's sweeps
        //                                and stage set-up (`items[i0].w`), all in front of the write-out's closing barrier
        //   shx[0..1]                    (MONO filter bounds) written with the records;  last read: sweep 1's head
        //   U / region A                 zeroed by this row's own threads in front of the closing barrier (MONO + LDS U: each thread the
        //                                entries it read), so the next row's bitmap finds zeros without a barrier of its own
        // select_fast (CLEAN) leaves SH_CNT2 / SH_EQ / SH_NHI at zero BEHIND its closing barrier for the same reason (the race of
        // commit 5b758b7 was exactly a reset in FRONT of a barrier that slower waves still read behind).
        // (no barrier here: every path of the next row's setup has one in front of its first use of the storage cleared above, and its
        // writes in front of that barrier — the counters of thread 0, the item records, the sort scratch — touch nothing this row's
        // tail still reads: the last reads of sh[] lie in front of a barrier of the write-out.  The one variant whose write-out has no
        // barrier behind its read of SH_CNT — general epilogue, U in global memory — keeps this one)
        if (!(U_LDS || MLIKE)) wg_sync<U_LDS>();
        PHASE_END(PH_OUTPUT);
    }
    if (timing) {
#pragma unroll
        for (int i = 0; i < PH_N; ++i) atomicAdd(&p.phase_cycles[i], ph[i]);
    }
#undef PHASE_END
}

}  // namespace
