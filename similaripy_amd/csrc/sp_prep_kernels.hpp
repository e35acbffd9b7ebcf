// sp_prep_kernels.hpp — the small launches in front of the row kernels: work per row, work-ordered queue,
// classified row descriptors, column-term minima, column term folded into the m2 stream.
#pragma once
#include "sp_common.hpp"

namespace {


// ---- work-ordered row queue (longest-processing-time-first, to within a factor 2) ----
// Rows are visited in descending MACs(t) buckets (bucket = floor(log2(work))), so that one huge row at the end of
// the target list cannot become the tail of the launch on skewed (power-law) matrices.
// One wave per row, grid-stride: a workgroup keeps its bucket histogram in LDS and adds it to the global one ONCE at
// the end (a single global word only sustains ~88 atomics/us; one atomic per row-block cost 2.8 ms on 1M rows).
// Rows of more than ROW_WORK_LONG entries (a popular item of a ratings matrix: 10^5 - 10^6) are only LISTED here (long_list, counted in
// *long_count) and summed by sp_row_work_long_kernel, a workgroup per row: one wave walking 800 k entries was 1.3 ms at the end of a
// 10 k-row call — the per-launch constant that kept the MovieLens-32M shape at x4.75 on eight slices (round 3, "Open").
constexpr int ROW_WORK_LONG = 8192;
__global__ __launch_bounds__(1024) void sp_row_work_kernel(int n_targets, const int *targets, const int *m1_indices,
                                                            const int *m1_indptr, const int *m2_indptr, unsigned *work,
                                                            unsigned *bucket_count, int *long_list, unsigned *long_count) {
    __shared__ unsigned hist[32];
    if (threadIdx.x < 32) hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int waves_total = (int)(gridDim.x * (blockDim.x >> 6));
    for (int gw = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); gw < n_targets; gw += waves_total) {
        const int t = targets[gw];
        const int s = m1_indptr[t], e = m1_indptr[t + 1];
        if (e - s > ROW_WORK_LONG) {          // (wave-uniform)
            if (lane == 0) long_list[atomicAdd(long_count, 1u)] = gw;
            continue;
        }
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

__global__ __launch_bounds__(1024) void sp_row_work_long_kernel(const int *targets, const int *m1_indices, const int *m1_indptr, const int *m2_indptr,
                                                                 unsigned *work, unsigned *bucket_count, const int *long_list, const unsigned *long_count) {
    __shared__ u64 part[16];
    const int n_long = (int)*long_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = blockIdx.x; i < n_long; i += gridDim.x) {
        const int gw = long_list[i];
        const int t = targets[gw];
        const int s = m1_indptr[t], e = m1_indptr[t + 1];
        u64 acc = 0;
        for (int j = s + (int)threadIdx.x; j < e; j += 1024) {
            const int u = m1_indices[j];
            acc += (u64)(m2_indptr[u + 1] - m2_indptr[u]);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
        __syncthreads();
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 tot = 0;
            for (int w = 0; w < 16; ++w) tot += part[w];
            const unsigned w32 = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)tot;
            work[gw] = w32;
            atomicAdd(&bucket_count[31 - __clz((int)(w32 | 1u))], 1u);
        }
    }
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

// Row descriptors, classified: what a kernel needs to start a row, one 32-byte record per queue position, so that
// its dependent-load chain is queue -> descriptor -> m1 entries -> m2 row bounds.  Rows the sparse kernel can
// take (few expected column collisions, few m1 entries) go to the sparse queue, everything else to the generic
// queue; positions are claimed with one atomic per wave and class, which keeps the (descending work) order of
// the input up to wave granularity.
struct ClassifyParams {
    int sparse_path;       // 0: everything is generic
    int n_cols, T;
    int nb_log2;           // sparse bitmap bits
    int cs_slots;          // collision-set slots of the sparse kernel
    int duo;               // its two-per-CU shape runs (aliasing 2^19-bit bitmap, half of the slots addressed by rank)
    int duo_l;             // ... and a second launch in the larger layout takes the rows between the two limits (the wave kernel's queue)
    int mono;              // the monotone sparse kernel runs: descriptor word 1.y carries den (val = xy / den), rows
    int any_norm;          //   with a normalised epilogue need den > 0 to be sparse
    float l2, l3;
    // heavy generic rows are queued as pieces (ranges of fine column windows, see sp_m2_splits_kernel) instead of one entry
    int split_fine;            // fine windows per row (0 = off)
    int split_pmax;            // pieces per row at most (the merge buffer holds split_pmax * k records)
    unsigned split_macs;       // a row gets one piece per this many MACs (and is split at all from twice that on)
    int split_cap;             // at most this many rows
    int *split_count;          // [2] rows split so far, pieces handed out so far (zero on entry)
    // light sparse rows have a kernel (and a queue) of their own: one wave per row (sp_wave_kernel.hpp)
    int wave;                  // 1 = that kernel runs in this call
    unsigned wave_macs_max;    // a sparse row goes to it with at most this many MACs and at most 64 m1 entries (its trips come from the prepass)
    unsigned *qcount_w;        // rows in its queue
    int4 *desc_w;              // its queue
    int4 *split_rows;          // [split_cap] {output slot, first piece, pieces, 0}
    int2 *piece_info;          // [split_cap * split_pmax] {output slot, first fine window | one past the last << 16}
};

__global__ __launch_bounds__(256) void sp_row_desc_kernel(int n_targets, const int *targets, const int *m1_indptr, const unsigned *work,
                                                           const unsigned *ordered_flag, const int *order, const float *Xtv,
                                                           const float *Xcos, const float *Xdep, ClassifyParams cp,
                                                           unsigned *qcount, int4 *desc_s, int4 *desc_g) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = pos < n_targets;
    int4 d0 = make_int4(0, 0, 0, 0), d1 = d0;
    bool sparse = false, wavey = false, duo_l_row = false;
    if (valid) {
        const int slot = (ordered_flag != nullptr && ordered_flag[0] != 0u) ? order[pos] : pos;
        const int t = targets[slot];
        const int s = m1_indptr[t], e = m1_indptr[t + 1];
        const unsigned macs = work[slot];
        d0 = make_int4(slot, t, s, e - s);
        d1 = make_int4((int)macs, Xtv ? (int)__float_as_uint(Xtv[t]) : 0, Xcos ? (int)__float_as_uint(Xcos[t]) : 0,
                       Xdep ? (int)__float_as_uint(Xdep[t]) : 0);
        bool den_ok = true;
        if (cp.mono) {
            // the epilogue's denominator with the column term folded away (s_plus.h:134-150): l2 * Xcos[t] or l3 * Xdep[t]
            float den = 1.f;
            if (cp.any_norm) den = Xcos ? cp.l2 * (Xcos[t] * 1.f) : (Xdep ? cp.l3 * (Xdep[t] * 1.f) : 0.f);
            d1.y = (int)__float_as_uint(den);
            den_ok = !cp.any_norm || den > 0.f;
        }
        if (cp.sparse_path && den_ok && macs > 0u && macs < (1u << 30) && (e - s) <= SORT_MAX && cp.n_cols > cp.T) {
            // expected number of products that find their bit set: true collisions + bitmap aliasing
            const float m = (float)macs;
            const float alias = (cp.nb_log2 < 31 && (1 << cp.nb_log2) < cp.n_cols) ? 1.f / (float)(1 << cp.nb_log2) : 0.f;
            const float expect = 0.5f * m * m * (1.f / (float)cp.n_cols + alias);
            sparse = expect <= 0.30f * (float)cp.cs_slots;
            // DUO: products that find their bit set = pairs of products on one bit, MACs^2 / (2 * bits) (a column pair on one bit counts
            // like a column that repeats); they must fit the rank-addressed half of the set — 0.44 x 4096 = 1 802 of 2 048 (a C2 row:
            // 1 600 +- 40; a row that does not fit after all is handed to the generic queue by the kernel)
            if (cp.duo) {
                const float est = 0.5f * m * m / (float)min(cp.n_cols, 1 << cp.nb_log2);
                sparse = est <= 0.44f * (float)cp.cs_slots;
                if (!sparse && cp.duo_l && est <= 0.88f * (float)DUO_CS_DIRECT_L) { sparse = true; duo_l_row = true; }
            }
        }
        wavey = (sparse && cp.wave && (e - s) <= 64 && macs <= cp.wave_macs_max) || duo_l_row;
        if (wavey && !duo_l_row) {
            // the wave kernel's collision set has 384 rank-addressed slots (sp_wave_kernel.hpp WV_CS_DIRECT): the row's expected marks must fit
            // with room to spare (307 is also what the 256-thread shape's own rule admits).  Its column bitmap has 2^17 bits: beyond,
            // aliased columns are marked like true collisions (and the bar is lower: 230).
            const float m = (float)macs;
            const bool alias = cp.n_cols > (1 << 17);
            wavey = 0.5f * m * m * (1.f / (float)cp.n_cols + (alias ? 1.f / (float)(1 << 17) : 0.f)) <= (alias ? 0.6f : 0.8f) * 384.f;
        }
    }
    // heavy generic rows: one queue entry per piece
    int split_id = -1, n_pieces = 0, per_piece = 0, piece0 = 0;
    if (valid && !sparse && cp.split_fine > 1 && (unsigned)d1.x >= 2u * cp.split_macs) {
        const int wanted = (int)min((unsigned)cp.split_pmax, ((unsigned)d1.x + cp.split_macs - 1u) / cp.split_macs);
        per_piece = (cp.split_fine + wanted - 1) / wanted;                  // fine windows per piece
        n_pieces = (cp.split_fine + per_piece - 1) / per_piece;
        if (n_pieces > 1) {
            const int id = atomicAdd(&cp.split_count[0], 1);
            if (id < cp.split_cap) {
                split_id = id;
                piece0 = atomicAdd(&cp.split_count[1], n_pieces);           // (never beyond split_cap * split_pmax: id < split_cap)
                cp.split_rows[id] = make_int4(d0.x, piece0, n_pieces, 0);
            }
        }
    }
    const int n_g = (valid && !sparse) ? (split_id >= 0 ? n_pieces : 1) : 0;     // generic queue entries of this lane
    const u64 ms = __ballot(valid && sparse && !wavey), mw = __ballot(valid && wavey);
    int incl = n_g;                                                                      // inclusive wave scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    const int tot_g = __shfl(incl, 63, 64);
    unsigned bs = 0, bg = 0, bw = 0;
    if (lane == 0) {
        if (mw) bw = atomicAdd(cp.qcount_w, (unsigned)__popcll(mw));
        if (ms) bs = atomicAdd(&qcount[0], (unsigned)__popcll(ms));
        if (tot_g) bg = atomicAdd(&qcount[1], (unsigned)tot_g);
    }
    bs = (unsigned)__builtin_amdgcn_readfirstlane((int)bs);
    bg = (unsigned)__builtin_amdgcn_readfirstlane((int)bg);
    bw = (unsigned)__builtin_amdgcn_readfirstlane((int)bw);
    if (valid) {
        const u64 below = (1ull << lane) - 1ull;
        if (wavey) {
            int4 *dst = cp.desc_w + 2 * (size_t)(bw + (unsigned)__popcll(mw & below));
            dst[0] = d0;
            dst[1] = d1;
        } else if (sparse) {
            int4 *dst = desc_s + 2 * (size_t)(bs + (unsigned)__popcll(ms & below));
            dst[0] = d0;
            dst[1] = d1;
        } else {
            int4 *dst = desc_g + 2 * (size_t)(bg + (unsigned)(incl - n_g));
            if (split_id < 0) { dst[0] = d0; dst[1] = d1; }
            else {
                for (int j = 0; j < n_pieces; ++j) {
                    const int piece = piece0 + j;
                    const int g0 = j * per_piece, g1 = min(cp.split_fine, g0 + per_piece);
                    cp.piece_info[piece] = make_int2(d0.x, g0 | (g1 << 16));
                    int4 e0 = d0;
                    e0.x = -1 - piece;
                    dst[2 * j] = e0;
                    dst[2 * j + 1] = d1;
                }
            }
        }
    }
}

// Work items ("trips") of the rows in the sparse queue, once per call (the sparse kernel's row setup — segment order, item and
// flat-start prefixes, the item records — used to be ~4.7 k of a C2 row's 74 k cycles, done by one wave while fifteen waited).
// One wave per row: segment i = m1 entry i, visited in descending |m1 value| (each segment scales its m2 row by its m1 value: the
// large products come first and the running k-th value starts high).
//
// A TRIP is what one wave fetches with one 16-byte load per lane: 64 lanes x 4 consecutive elements.  Cutting every m2 row into
// its own trips leaves the last trip of each row partly empty (C2: rows of ~640 elements = 2.5 trips, a sixth of all lanes idle;
// a user-scoring row of 100 elements fills 25 lanes of 64) — and the sweeps' time goes with the NUMBER of trips (each is a memory
// round trip for its wave), not with the bytes.  So the segments are laid end to end on a virtual lane axis (a segment of `len`
// elements takes ceil(len/4) lanes) and a trip is a window of 64 lanes of that axis: the lanes [0, sB) continue (or start) one
// segment — piece A — and the lanes [sB, 64) start the next one — piece B.  At most TWO pieces per trip: a segment that would be
// the third one in a window starts at the next window instead (only segments shorter than 64 lanes ever cause that).
// Records (16 B each), the image of the kernel's LDS item area:
//   [T] for T < n_trips:  {byte offset of A's first element in m2, elements of A in this trip, m1 value bits of A,
//                          products before this trip | (index of the B record << 20, 0 = no B)}
//   [n_trips]             the sentinel {OOB_SOFFSET, 0, 0, all products}
//   [n_trips + 1 ...]     B records {byte offset, elements, m1 value bits, first lane sB}
// row[0] = {first index, length} of the row's MATRIX-filter list (0, 0 without one); n_trips and n_records go into the row's queue descriptor (desc_n_trips / desc_n_rec of its .w); rows of more
// than 64 entries, more than ITEMS_PRE records or 2^20 products keep 0 / 0 there: the kernel sets those up itself (one piece per trip).  pack = 0: one piece per trip for every row (the 1024-thread shape).
__global__ __launch_bounds__(256) void sp_row_items_kernel(const unsigned *__restrict__ qcount, int items_rows, int4 *__restrict__ desc_s,
                                                            const int *__restrict__ m1_indices, const float *__restrict__ m1_data,
                                                            const int *__restrict__ m2_indptr, int4 *__restrict__ items_g, int pack,
                                                            const int *__restrict__ f_indptr, int stride) {
    // the row's records are put together in LDS and leave in 1 KB stores (lane l: records l, l + 64, ...): written in place, every lane's two or
    // three records of a segment went out at a 48-byte stride — three partial stores per 128-byte line (C2: 1.69 -> 1.42 ms per step; the kernel
    // is bound by its 3 GB of record writes: a cheaper ranking and a three-row load pipeline were measured at 0.0 ms, profiles/r05_exp_dropped.txt)
    __shared__ int4 img_all[4][ITEMS_STRIDE];
    int4 *img = img_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const int n_rows = (int)qcount[0];
    const int waves_total = (int)(gridDim.x * (blockDim.x >> 6));
    for (int q = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); q < n_rows; q += waves_total) {
        const int4 d = desc_s[2 * (size_t)q];
        const int slot = __builtin_amdgcn_readfirstlane(d.x), s = __builtin_amdgcn_readfirstlane(d.z), n1 = __builtin_amdgcn_readfirstlane(d.w);
        const int t_row = __builtin_amdgcn_readfirstlane(d.y);
        if (slot >= items_rows) continue;
        int4 *row = items_g + (size_t)slot * (size_t)stride;
        if (n1 > 64) continue;
        int r0_in = 0, len_in = 0;
        unsigned vbits_in = 0u;
        if (lane < n1) {
            const int u = m1_indices[s + lane];
            vbits_in = __float_as_uint(m1_data[s + lane]);
            r0_in = m2_indptr[u];
            len_in = m2_indptr[u + 1] - r0_in;
        }
        const unsigned key = (lane < n1 && len_in > 0) ? ((vbits_in & 0x7FFFFFFFu) | 1u) : 0u;      // 0 = no segment
        // position of this lane's segment in descending key order (ties: lower lane first; empty lanes last): a permutation
        int rank = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, j);
            rank += (kj > key || (kj == key && j < lane)) ? 1 : 0;
        }
        // from here on in POSITION order: lane i holds the i-th segment visited
        const int r0 = __builtin_amdgcn_ds_permute(rank * 4, key != 0u ? r0_in : 0);
        const int len = __builtin_amdgcn_ds_permute(rank * 4, key != 0u ? len_in : 0);
        const unsigned vbits = (unsigned)__builtin_amdgcn_ds_permute(rank * 4, key != 0u ? (int)vbits_in : 0);
        const int n_seg = __popcll(__ballot(key != 0u));
        const int L = (len + 3) >> 2;                                   // lanes of the virtual axis
        const int fs_incl = wave_incl_scan_dpp(len);
        const int E = fs_incl - len;                                    // products of the segments before this one
        const int total = __builtin_amdgcn_readlane(fs_incl, 63);
        // first virtual lane of every segment: a sequential walk in scalar registers (<= 64 steps)
        int V = 0, cur = 0, last_b = -1;
        for (int j = 0; pack && j < n_seg; ++j) {
            const int Lj = __builtin_amdgcn_readlane(L, j);
            int at = cur;
            if ((at & 63) != 0 && last_b == (at >> 6)) at = (at + 63) & ~63;      // that window has its second piece already
            if ((at & 63) != 0) last_b = at >> 6;
            if (lane == j) V = at;
            cur = at + Lj;
        }
        // Packing costs the sweeps ~25 instructions per trip (per-lane offsets, counts and m1 values instead of scalars): it pays
        // when it removes a quarter of the trips or more (user-scoring rows of ~100 elements: half of them); otherwise (C2: rows of
        // ~640 elements, 14 % fewer trips, no gain measured) every segment starts its own window, i.e. one piece per trip
        {
            const int t_incl = wave_incl_scan_dpp((L + 63) >> 6);
            const int unpacked = __builtin_amdgcn_readlane(t_incl, 63);
            if (!pack || 4 * ((cur + 63) >> 6) > 3 * unpacked) {
                V = (t_incl - ((L + 63) >> 6)) * 64;
                cur = unpacked * 64;
            }
        }
        const int n_trips = (cur + 63) >> 6;
        // the segment behind this one (it may be this segment's piece B in this segment's last window)
        const int nx = ((lane + 1) & 63) * 4;
        const int Vn = __builtin_amdgcn_ds_bpermute(nx, V), Ln = __builtin_amdgcn_ds_bpermute(nx, L);
        const int r0n = __builtin_amdgcn_ds_bpermute(nx, r0), lenn = __builtin_amdgcn_ds_bpermute(nx, len);
        const int vbn = __builtin_amdgcn_ds_bpermute(nx, (int)vbits);
        const bool mine = lane < n_seg;
        const bool has_b = mine && (lane + 1 < n_seg) && (Vn & 63) != 0;
        const int b_incl = wave_incl_scan_dpp(has_b ? 1 : 0);
        const int n_b = __builtin_amdgcn_readlane(b_incl, 63);
        const int n_rec = n_trips + 1 + n_b;
        if (n_seg == 0 || n_rec > stride - 1 || n_trips > 0x3FF || total >= (1 << 20)) continue;      // (more records than the call's stride holds: set up in the kernel)
        if (mine) {
            const int bidx = n_trips + 1 + (b_incl - 1);                // (only read when has_b)
            const int t_first = (V >> 6) + ((V & 63) != 0 ? 1 : 0);     // first window whose lane 0 lies inside this segment
            const int t_last = (V + L - 1) >> 6;
            for (int T = t_first; T <= t_last; ++T) {
                const int a = 64 * T - V;                               // lanes of the segment before this window
                const int cnt = min(4 * min(64, L - a), len - 4 * a);
                img[1 + T] = make_int4((r0 + 4 * a) * 4, cnt, (int)vbits, (E + 4 * a) | ((T == t_last && has_b) ? (bidx << 20) : 0));
            }
            if (has_b) {
                const int sb = Vn & 63;
                img[1 + bidx] = make_int4(r0n * 4, min(4 * min(64 - sb, Ln), lenn), vbn, sb);
            }
        }
        if (lane == 0) {
            // record 0: where the row's MATRIX-filter list starts and how long it is (monotone variant: the row kernel then has
            // the list's columns on their way before its first sweep)
            int f0 = 0, fl = 0;
            if (f_indptr != nullptr) { f0 = f_indptr[t_row]; fl = f_indptr[t_row + 1] - f0; }
            img[0] = make_int4(f0, fl, 0, 0);
            img[1 + n_trips] = make_int4((int)OOB_SOFFSET, 0, 0, total);
            // the counts travel in the row's descriptor (the row kernel has it in scalar registers two rows ahead)
            ((int *)&desc_s[2 * (size_t)q])[3] = n1 | (n_trips << 9) | (n_rec << 19);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (one wave owns the image: LDS runs its accesses in order)
        for (int i = lane; i <= n_rec; i += 64) row[i] = img[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// Work records of the rows in the WAVE queue (sp_wave_kernel.hpp), once per call: one 12-byte record per SEGMENT, 64 per row (768 B).
// The wave kernel lays a row's segments end to end on the virtual lane axis with NO gaps — any number of pieces per trip (the
// two-piece rule above leaves a user-scoring row of 100-element segments at 50 lanes of 64 per trip) — and finds every lane's segment
// by counting segment starts (v_mbcnt on a per-trip start mask), so a record only has to say where the segment's elements are, how
// many, and what they are scaled by: record i (position order: descending |m1 value|) = {4 r0 = byte offset of the m2 row, len, m1 value
// bits}; the kernel's scan of ceil(len / 4) gives the segment's first lane V, and with B = 4 r0 - 16 V, D = len + 4 V
//     byte offset into m2  = B + 16 u        elements left for the lane = D - 4 u        (u = 64 T + lane, the lane's place on the axis)
// hold for every lane of the segment.  Records behind the last segment are zero.  n_trips = ceil(lanes / 64) and the segment count go
// into the row's queue descriptor; rows of more than 64 entries, more than 63 trips or 2^20 products keep 0 trips there and the wave
// kernel hands them to the generic kernel.
struct __attribute__((packed, aligned(4))) WaveSegRec { int off4; int len; unsigned vbits; };
constexpr int WAVE_ITEMS_STRIDE = 64 * (int)sizeof(WaveSegRec) / 16;      // 48 sixteen-byte units per output slot
__global__ __launch_bounds__(256) void sp_row_items_wave_kernel(const unsigned *__restrict__ qcount, int items_rows, int4 *__restrict__ desc_w,
                                                                 const int *__restrict__ m1_indices, const float *__restrict__ m1_data,
                                                                 const int *__restrict__ m2_indptr, int4 *__restrict__ items_g, int stride) {
    const int lane = threadIdx.x & 63;
    const int n_rows = (int)qcount[0];
    const int waves_total = (int)(gridDim.x * (blockDim.x >> 6));
    for (int q = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); q < n_rows; q += waves_total) {
        const int4 d = desc_w[2 * (size_t)q];
        const int slot = __builtin_amdgcn_readfirstlane(d.x), s = __builtin_amdgcn_readfirstlane(d.z), n1 = __builtin_amdgcn_readfirstlane(d.w);
        if (slot >= items_rows || n1 > 64 || stride < WAVE_ITEMS_STRIDE) continue;
        int r0_in = 0, len_in = 0;
        unsigned vbits_in = 0u;
        if (lane < n1) {
            const int u = m1_indices[s + lane];
            vbits_in = __float_as_uint(m1_data[s + lane]);
            r0_in = m2_indptr[u];
            len_in = m2_indptr[u + 1] - r0_in;
        }
        // position in descending |m1 value| order, empty lanes last.  The order is a heuristic (large products first: the running k-th value
        // starts high), so the key keeps the value's top 24 bits and ends in 63 - lane: all 64 keys differ and a position is a plain count
        // (a readlane, a compare and an add-with-carry per lane; the exact order with its tie rule costs five)
        const bool real = lane < n1 && len_in > 0;
        const unsigned key = (real ? (0x80000000u | (((vbits_in & 0x7FFFFFFFu) >> 7) << 6)) : 0u) | (unsigned)(63 - lane);
        int rank = 0;
        for (int j = 0; j < 64; ++j) {
            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)key, j);
            rank += (kj > key) ? 1 : 0;
        }
        const int r0 = __builtin_amdgcn_ds_permute(rank * 4, real ? r0_in : 0);
        const int len = __builtin_amdgcn_ds_permute(rank * 4, real ? len_in : 0);
        const unsigned vbits = (unsigned)__builtin_amdgcn_ds_permute(rank * 4, real ? (int)vbits_in : 0);
        const int n_seg = __popcll(__ballot(real));
        const int L = (len + 3) >> 2;
        const int l_incl = wave_incl_scan_dpp(L);
        const int lanes = __builtin_amdgcn_readlane(l_incl, 63);
        const int total = __builtin_amdgcn_readlane(wave_incl_scan_dpp(len), 63);
        const int n_trips = (lanes + 63) >> 6;
        if (n_seg == 0 || n_trips > 63 || total >= (1 << 20)) continue;
        WaveSegRec *row = (WaveSegRec *)(items_g + (size_t)slot * (size_t)stride);
        row[lane] = (lane < n_seg) ? WaveSegRec{(int)(4u * (unsigned)r0), len, vbits} : WaveSegRec{0, 0, 0u};
        if (lane == 0) ((int *)&desc_w[2 * (size_t)q])[3] = n1 | (n_trips << 9) | (n_seg << 19);
    }
}

// The top-k of a split row from its pieces' results (each a top-k of its own column window, threshold applied):
// ascending sort of the pieces' {value key, column} records in LDS, the k largest go to the row's output slot.
// One workgroup per split row; at most split_pmax * k <= 8192 records.
constexpr int MERGE_NT = 1024;      // (round 6: 256 threads took 152 us per split row — a latency chain of ~90 barrier steps, sixteen trips each)
__global__ __launch_bounds__(MERGE_NT) void sp_merge_pieces_kernel(const int *__restrict__ split_count, int split_cap, const int4 *__restrict__ split_rows, int k,
                                                               const int *__restrict__ targets, const int *__restrict__ part_cols, const float *__restrict__ part_vals,
                                                               const int *__restrict__ part_counts, int *__restrict__ rows, int *__restrict__ cols,
                                                               float *__restrict__ values, int *__restrict__ counts) {
    extern __shared__ unsigned long long mg_buf[];
    __shared__ int mg_n;
    const int n_split = min(*split_count, split_cap);
    const int tid = threadIdx.x;
    for (int s = blockIdx.x; s < n_split; s += gridDim.x) {
        if (tid == 0) mg_n = 0;
        __syncthreads();
        const int4 sr = split_rows[s];
        for (int j = 0; j < sr.z; ++j) {
            const int piece = sr.y + j;
            const int n = part_counts[piece];
            int base = 0;
            if (tid == 0) { base = mg_n; mg_n = base + n; }
            __syncthreads();
            base = mg_n - n;
            for (int i = tid; i < n; i += MERGE_NT)
                mg_buf[base + i] = ((unsigned long long)fkey(part_vals[(size_t)piece * k + i]) << 32) | (unsigned long long)(unsigned)part_cols[(size_t)piece * k + i];
            __syncthreads();
        }
        const int n = mg_n;
        // ascending bitonic network with virtual padding (the transpose's, restated for this buffer)
        if (n > 1) {
            int lp = 1;
            while ((1 << lp) < n) ++lp;
            const int half = 1 << (lp - 1);
            for (int lk = 1; lk <= lp; ++lk) {
                const int kk = 1 << lk, hk = kk >> 1;
                for (int t = tid; t < half; t += MERGE_NT) {
                    const int blk = t >> (lk - 1), off = t & (hk - 1);
                    const int lo = (blk << lk) + off, hi = (blk << lk) + (kk - 1 - off);
                    if (hi < n) { const unsigned long long a = mg_buf[lo], c = mg_buf[hi]; if (a > c) { mg_buf[lo] = c; mg_buf[hi] = a; } }
                }
                __syncthreads();
                for (int j = kk >> 2; j > 0; j >>= 1) {
                    for (int t = tid; t < half; t += MERGE_NT) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                        if (hi < n) { const unsigned long long a = mg_buf[lo], c = mg_buf[hi]; if (a > c) { mg_buf[lo] = c; mg_buf[hi] = a; } }
                    }
                    __syncthreads();
                }
            }
        }
        const int slot = sr.x;
        const int n_out = min(n, k);
        const long long o = (long long)slot * (long long)k;
        for (int j = tid; j < k; j += MERGE_NT) {
            int r = 0, c = 0;
            float v = 0.f;
            if (j < n_out) {
                const unsigned long long it = mg_buf[n - 1 - j];
                r = targets[slot];
                c = (int)(unsigned)(it & 0xFFFFFFFFull);
                v = funkey((unsigned)(it >> 32));
            }
            if (rows) rows[o + j] = r;
            cols[o + j] = c;
            values[o + j] = v;
        }
        if (tid == 0 && counts) counts[slot] = n_out;
        __syncthreads();
    }
}

// Minima of the three column-term vectors over all columns (feeds Epi::upper).  Many workgroups (one walked 10^6 columns in 0.57 ms): a
// workgroup's minima go into out[] as INVERTED order keys by atomicMax — out[] is part of the zeroed workspace header, and 0 is "nothing yet" —
// and the workgroup that arrives last (counter `done`, zeroed with the header) turns the keys back into floats.
__device__ __forceinline__ unsigned min_key(float m) { return ~fkey(m); }            // larger key = smaller value; never 0 for a finite or infinite float
__device__ __forceinline__ float min_unkey(unsigned k) { return funkey(~k); }
__global__ __launch_bounds__(1024) void sp_colterm_min_kernel(int n_cols, const float *Ytv, const float *Ycos, const float *Ydep, float *out, unsigned *done) {
    __shared__ float red[3][16];
    __shared__ bool last;
    const float inf = __builtin_inff();
    float m0 = inf, m1 = inf, m2 = inf;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n_cols; i += (long long)gridDim.x * 1024) {
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
        if (m < inf) atomicMax((unsigned *)out + threadIdx.x, min_key(m));
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(done, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (last && threadIdx.x < 3) {
        __threadfence();
        const unsigned k = atomicMax((unsigned *)out + threadIdx.x, 0u);      // (a read that every earlier atomic is ordered before)
        out[threadIdx.x] = k ? min_unkey(k) : 0.f;   // vector not in use (or empty): its weight is 0 anyway
    }
}

// ---- per-call passes of the sparse kernel's BOUNDED variant (MODE 2; BndInfo in sp_common.hpp) ----
// a column is VALID when every live Y_j is finite and >= 0 and its W is a finite normal float well above the bottom (code >= 1)
__device__ __forceinline__ bool bnd_valid(float ytv, float ycos, float ydep, float w, int id_bits) {
    const float inf = __builtin_inff();
    return (ytv >= 0.f) && (ycos >= 0.f) && (ydep >= 0.f) && (ytv < inf) && (ycos < inf) && (ydep < inf) && (w < inf) && ((__float_as_uint(w) >> (id_bits - 1)) >= 2u) &&
           !(__float_as_uint(w) >> 31);
}
// (1) reference multipliers rho_j (l_j x the mean of the row terms that are finite and positive) and the minima of the live Y_j over the
//     valid columns -> BndInfo.  Two launches of many workgroups (one workgroup walked 10^6 rows and 10^6 columns in 0.92 ms): the row
//     means first — partial sums by float atomics into the zeroed header words `acc` {sum cos, sum dep, n cos, n dep, done}, the workgroup
//     that arrives last writes rho —, then the columns: the count of valid columns adds up in BndInfo::state and the minima collect as
//     inverted order keys in the ymin fields (all zero at the start: the header is), the last workgroup makes them the final BndInfo.
__global__ __launch_bounds__(1024) void sp_bnd_xmean_kernel(int n_rows_m1, const float *Xcos, const float *Xdep, bool has_tv, bool has_cos, bool has_dep,
                                                             float l1t2, float l2, float l3, float *acc, BndInfo *out) {
    __shared__ double redd[2][16];
    __shared__ unsigned redc[2][16];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s0 = 0.0, s1 = 0.0;
    unsigned c0 = 0, c1 = 0;
    for (long long i = (long long)blockIdx.x * 1024 + tid; i < n_rows_m1; i += (long long)gridDim.x * 1024) {
        if (Xcos) { const float x = Xcos[i]; if (x > 0.f && x < __builtin_inff()) { s0 += (double)x; ++c0; } }
        if (Xdep) { const float x = Xdep[i]; if (x > 0.f && x < __builtin_inff()) { s1 += (double)x; ++c1; } }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64);
        c0 += __shfl_xor(c0, d, 64); c1 += __shfl_xor(c1, d, 64);
    }
    if (lane == 0) { redd[0][wave] = s0; redd[1][wave] = s1; redc[0][wave] = c0; redc[1][wave] = c1; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        unsigned na = 0, nb = 0;
        for (int w = 0; w < 16; ++w) { a += redd[0][w]; b += redd[1][w]; na += redc[0][w]; nb += redc[1][w]; }
        // (the mean only has to be the SAME number for every user of this BndInfo, not an exact one: float partial sums)
        if (na) { atomicAdd(&acc[0], (float)a); atomicAdd((unsigned *)&acc[2], na); }
        if (nb) { atomicAdd(&acc[1], (float)b); atomicAdd((unsigned *)&acc[3], nb); }
        __threadfence();
        last = atomicAdd((unsigned *)&acc[4], 1u) == gridDim.x - 1u;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        const float a = atomicAdd(&acc[0], 0.f), b = atomicAdd(&acc[1], 0.f);
        const unsigned na = atomicAdd((unsigned *)&acc[2], 0u), nb = atomicAdd((unsigned *)&acc[3], 0u);
        float rho[3];
        rho[0] = has_tv ? l1t2 : 0.f;
        rho[1] = (has_cos && na) ? l2 * (a / (float)na) : 0.f;
        rho[2] = (has_dep && nb) ? l3 * (b / (float)nb) : 0.f;
        for (int j = 0; j < 3; ++j) if (!(rho[j] > 0.f) || !(rho[j] < __builtin_inff())) rho[j] = 0.f;
        out->rho_tv = rho[0]; out->rho_cos = rho[1]; out->rho_dep = rho[2];
    }
}
__global__ __launch_bounds__(1024) void sp_bnd_range_kernel(int n_cols, const float *Ytv, const float *Ycos, const float *Ydep, unsigned *done, BndInfo *out, int id_bits) {
    __shared__ unsigned redc[16];
    __shared__ float redm[3][16];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float rtv = out->rho_tv, rcos = out->rho_cos, rdep = out->rho_dep;      // (written by the launch before this one)
    const float inf = __builtin_inff();
    unsigned nv = 0;
    float m0 = inf, m1 = inf, m2 = inf;
    for (long long i = (long long)blockIdx.x * 1024 + tid; i < n_cols; i += (long long)gridDim.x * 1024) {
        const float ytv = Ytv ? Ytv[i] : 0.f, ycos = Ycos ? Ycos[i] : 0.f, ydep = Ydep ? Ydep[i] : 0.f;
        if (bnd_valid(ytv, ycos, ydep, bnd_w(rtv, ytv, rcos, ycos, rdep, ydep), id_bits)) {
            ++nv;
            m0 = fminf(m0, ytv); m1 = fminf(m1, ycos); m2 = fminf(m2, ydep);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        nv += __shfl_xor(nv, d, 64);
        m0 = fminf(m0, __shfl_xor(m0, d, 64)); m1 = fminf(m1, __shfl_xor(m1, d, 64)); m2 = fminf(m2, __shfl_xor(m2, d, 64));
    }
    if (lane == 0) { redc[wave] = nv; redm[0][wave] = m0; redm[1][wave] = m1; redm[2][wave] = m2; }
    __syncthreads();
    if (tid == 0) {
        nv = 0;
        for (int w = 0; w < 16; ++w) { nv += redc[w]; m0 = fminf(m0, redm[0][w]); m1 = fminf(m1, redm[1][w]); m2 = fminf(m2, redm[2][w]); }
        if (nv) atomicAdd((unsigned *)&out->state, nv);
        if (m0 < inf) atomicMax((unsigned *)&out->ymin_tv, min_key(m0));
        if (m1 < inf) atomicMax((unsigned *)&out->ymin_cos, min_key(m1));
        if (m2 < inf) atomicMax((unsigned *)&out->ymin_dep, min_key(m2));
        __threadfence();
        last = atomicAdd(done, 1u) == gridDim.x - 1u;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        const unsigned n_valid = atomicAdd((unsigned *)&out->state, 0u);
        const unsigned k0 = atomicMax((unsigned *)&out->ymin_tv, 0u), k1 = atomicMax((unsigned *)&out->ymin_cos, 0u), k2 = atomicMax((unsigned *)&out->ymin_dep, 0u);
        out->ymin_tv = (Ytv && k0) ? min_unkey(k0) : 0.f;
        out->ymin_cos = (Ycos && k1) ? min_unkey(k1) : 0.f;
        out->ymin_dep = (Ydep && k2) ? min_unkey(k2) : 0.f;
        out->state = (n_valid > 0u && (rtv > 0.f || rcos > 0.f || rdep > 0.f)) ? 1 : 0;
    }
}

// (2) the packed id of every column: id | code << id_bits, code = (bits(W) >> (id_bits - 1)) - 1 (id_bits = 20 up to 2^20 columns, 21 / 22 beyond); a column that is not valid gets 0xFFFFFFFF (if an m2
//     entry points at one the pack pass below takes the whole call off the bounded variant)
__global__ __launch_bounds__(256) void sp_bnd_colpack_kernel(int n_cols, const float *__restrict__ Ytv, const float *__restrict__ Ycos, const float *__restrict__ Ydep,
                                                              const BndInfo *__restrict__ info, unsigned *__restrict__ colpack, int id_bits) {
    const BndInfo b = *info;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cols; i += gridDim.x * blockDim.x) {
        unsigned out = 0xFFFFFFFFu;
        if (b.state == 1) {
            const float ytv = Ytv ? Ytv[i] : 0.f, ycos = Ycos ? Ycos[i] : 0.f, ydep = Ydep ? Ydep[i] : 0.f;
            const float w = bnd_w(b.rho_tv, ytv, b.rho_cos, ycos, b.rho_dep, ydep);
            if (bnd_valid(ytv, ycos, ydep, w, id_bits)) out = (unsigned)i | (((__float_as_uint(w) >> (id_bits - 1)) - 1u) << id_bits);
        }
        colpack[i] = out;
    }
}

// (3) the m2 index stream with the codes in it.  An entry on an invalid column: state |= 2 (the general variant runs instead).
__global__ __launch_bounds__(256) void sp_bnd_pack_ids_kernel(long long nnz, const int *__restrict__ indices, const unsigned *__restrict__ colpack,
                                                               unsigned *__restrict__ out, BndInfo *__restrict__ info) {
    bool bad = false;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const int c = indices[i];
        const unsigned pk = colpack[c];
        bad |= pk == 0xFFFFFFFFu;
        out[i] = (pk == 0xFFFFFFFFu) ? (unsigned)c : pk;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&info->state, 2);
}

// Interleave the column terms in use into one {Ytv, Ycos, Ydep, 0} record per column (0 for a term whose weight is 0:
// the epilogue never looks at it), so that judging a candidate costs one gather.
__global__ __launch_bounds__(256) void sp_pack_colterms_kernel(int n_cols, const float *__restrict__ Ytv, const float *__restrict__ Ycos,
                                                                const float *__restrict__ Ydep, float4 *__restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cols; i += gridDim.x * blockDim.x)
        out[i] = make_float4(Ytv ? Ytv[i] : 0.f, Ycos ? Ycos[i] : 0.f, Ydep ? Ydep[i] : 0.f, 0.f);
}

// *flag |= 1 iff any value is negative (the Bayesian-shrink epilogue is not monotone in a negative raw dot)
__global__ __launch_bounds__(256) void sp_any_negative_kernel(long long nnz, const float *__restrict__ data, int *__restrict__ flag) {
    bool neg = false;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) neg |= data[i] < 0.f;
    if (__ballot(neg) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Fold the column term of a product-form epilogue into the m2 stream:  out[i] = data[i] / Y[indices[i]]
// (0 where Y is 0: the reference returns 0 for a zero denominator, s_plus.h:147-150).  One streaming pass.
// A STORED entry over a zero column term (a 'sum' depopularisation weight of signed data that cancels to exactly 0; a cosine term is 0
// for empty columns only): the reference reports such a column with value 0 whenever a product touches it; here every product on it is
// 0.0, the column is touched like any other (a sum of 0.0 is not "untouched": free slots are marked otherwise in every accumulator) and
// comes out with value 0 — the same.  *zero_term (optional) still reports that it happened (rounds 3-5 reran such calls unfolded).
__global__ __launch_bounds__(256) void sp_fold_colterm_kernel(long long nnz, const int *__restrict__ indices,
                                                               const float *__restrict__ data, const float *__restrict__ Y,
                                                               float *__restrict__ out, int *__restrict__ zero_term) {
    bool bad = false;
    // four entries per thread and trip (16-byte loads and stores: the pass ran at 2 TB/s with one — 0.47 ms of a configs[1] step);
    // the arrays start 16-byte aligned whenever they come from an allocator, anything else takes the scalar loop
    long long done = 0;
    if ((((size_t)indices | (size_t)data | (size_t)out) & 15u) == 0) {
        const long long n4 = nnz >> 2;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            const int4 c = ((const int4 *)indices)[i];
            const float4 d = ((const float4 *)data)[i];
            const float y0 = Y[c.x], y1 = Y[c.y], y2 = Y[c.z], y3 = Y[c.w];
            float4 o;
            o.x = (y0 != 0.f) ? d.x / y0 : 0.f;
            o.y = (y1 != 0.f) ? d.y / y1 : 0.f;
            o.z = (y2 != 0.f) ? d.z / y2 : 0.f;
            o.w = (y3 != 0.f) ? d.w / y3 : 0.f;
            ((float4 *)out)[i] = o;
            bad |= ((y0 == 0.f) && (d.x != 0.f)) || ((y1 == 0.f) && (d.y != 0.f)) || ((y2 == 0.f) && (d.z != 0.f)) || ((y3 == 0.f) && (d.w != 0.f));
        }
        done = n4 << 2;
    }
    for (long long i = done + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const float y = Y[indices[i]];
        const float d = data[i];
        out[i] = (y != 0.f) ? d / y : 0.f;
        bad |= (y == 0.f) && (d != 0.f);
    }
    if (zero_term != nullptr && bad) *zero_term = 1;
}

// Boundaries of the generic kernel's standard dense windows [j*width, (j+1)*width) inside every (sorted) m2 row:
// out[j*n_rows + u] (boundary-major) = first position of row u whose column id is >= (j+1)*width  (s_plus.h:385-394 does this lower_bound
// per target row and block; the boundaries do not depend on the target row)
// Launched BEHIND the sparse-row kernels, in front of the generic one: only the generic kernel reads the boundaries, and a call whose
// generic queue is empty by then (qcount_g: rows classified generic + the sparse kernels' give-ups) skips the pass — the headline shape
// paid 0.13 ms per step for a queue that is always empty (VERDICT r4 next #2a).  state[0] = 1 once the boundaries exist in this
// workspace (a SP_FLAG_REUSE_M2_PREP sub-launch whose predecessors all skipped builds them when it needs them), state[1] counts the
// workgroups that are done (the last one publishes).
constexpr unsigned SPLITS_MIN_QUEUE = 64u;      // generic-queue entries from which the window-boundary table is worth its pass over m2
__global__ __launch_bounds__(256) void sp_m2_splits_kernel(int n_rows, const int *__restrict__ indptr, const int *__restrict__ indices, int width, int n_splits,
                                                            int *__restrict__ out, const unsigned *__restrict__ qcount_g, int *__restrict__ state) {
    // (a generic queue of a handful of rows — configs[4]: 12 of 10^6 — does not pay for a pass over m2: its rows find their slices by lower_bound)
    if (*(volatile const unsigned *)qcount_g < SPLITS_MIN_QUEUE || *(volatile const int *)&state[0] != 0) return;      // (uniform over the grid: nothing in this launch writes either)
    // One WAVE per m2 row, one coalesced pass over its (ascending) column ids: element i starts window idx[i] / width; every window that
    // begins between element i - 1 and element i has its boundary at position i (round 4 ran a lower_bound — eight dependent loads — per
    // row and boundary: 0.27 ms per call at the MovieLens shape with 10 boundaries per row, twice that with the 20 of round 5's finer pieces)
    // The table is boundary-major (out[w * n_rows + u]): a heavy row's entries are ascending m2 rows, so the generic kernel's gathers of ONE boundary
    // for them lie next to each other (row-major they were 4 bytes out of every 80).  A workgroup takes 64 consecutive rows, sixteen per wave, collects
    // their boundaries in LDS and writes 256-byte lines.
    __shared__ int tile[32][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_tiles = ((long long)n_rows + 63) >> 6;
    // (round 6: the pass took 0.37 ms of a 7.4 ms N = 8 slice of configs[3] — a wave walked its sixteen rows one behind the other, two dependent
    // row-pointer loads in front of each, and divided every column id by the window width twice: the width is a power of two (2T / f) — a shift —,
    // and the wave's seventeen row pointers come with ONE load)
    const bool pow2 = (width & (width - 1)) == 0;
    const int wshift = pow2 ? 31 - __builtin_clz((unsigned)max(width, 1)) : 0;
    auto win = [&](int col) { return min(n_splits, pow2 ? (col >> wshift) : col / width); };
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long ub = t * 64 + wave * 16;
        const int my_ptr = (lane <= 16 && ub + lane <= n_rows) ? indptr[ub + lane] : 0;
        for (int q = 0; q < 16; ++q) {
            const int rt = wave * 16 + q;                      // row of the tile
            const long long u = ub + q;
            if (u >= n_rows) break;                            // (uniform per wave)
            const int r0 = __builtin_amdgcn_readlane(my_ptr, q), r1 = __builtin_amdgcn_readlane(my_ptr, q + 1);
            for (int i = r0 + lane; i <= r1; i += 64) {
                const int wprev = (i == r0) ? 0 : win(indices[i - 1]);
                const int wcur = (i == r1) ? n_splits : win(indices[i]);
                for (int w = wprev; w < wcur; ++w) tile[w][rt] = i;      // boundary w: the first position whose column id is >= (w + 1) * width
            }
        }
        __syncthreads();
        for (int w = wave; w < n_splits; w += 4) {
            const long long u = t * 64 + lane;
            if (u < n_rows) out[(size_t)w * (size_t)n_rows + (size_t)u] = tile[w][lane];
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&state[1], 1) == (int)gridDim.x - 1) { state[1] = 0; __threadfence(); atomicExch(&state[0], 1); }
    }
}

}  // namespace
