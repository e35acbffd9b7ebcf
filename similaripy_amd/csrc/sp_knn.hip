// sp_knn.hip — MI355X (gfx950 / CDNA4) top-k sparse row similarity: host side and C ABI (include/sp_knn.h).
//
// Replaces the reference's only native hot path,
//   s_plus::compute_similarities_parallel<int,float>   (similaripy/cython_code/s_plus.h:265-453)
// i.e. for every target row t of CSR m1:
//   acc[c] = sum_u m1[t,u] * m2[u,c]            (Gustavson row-wise SpGEMM, s_plus.h:418-438)
//   val[c] = epilogue(acc[c], X*[t], Y*[c])     (s_plus.h:129-156)
//   keep the k largest val[c] >= threshold that pass the column selectors (s_plus.h:192-215, 39-64)
//
// Launch sequence of one call (all on the caller's stream):
//   sp_fold_colterm_kernel | sp_pack_colterms_kernel, sp_colterm_min_kernel   column term folded into the m2 stream | column terms interleaved, their minima
//   sp_row_work_kernel, sp_bucket_base_kernel, sp_row_order_kernel   MACs per row, descending-work queue
//   sp_row_desc_kernel       classified 32-byte row descriptors: sparse queue / generic queue
//   sp_knn_sparse_kernel     (sp_sparse_kernel.hpp)  bitmap + two sweeps, the headline shape; persistent
//                            workgroups, one per CU, rows pulled from an atomic queue — the analogue of
//                            `omp for schedule(dynamic)` (s_plus.h:337); give-ups join the generic queue
//   sp_knn_generic_kernel    (sp_generic_kernel.hpp) LDS accumulator tile + column windows
// HBM-bound integer/float streaming work: no MFMA on purpose.  Everything is written for gfx950 only
// (wave64, 160 KiB LDS); there is no CPU path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/sp_knn.h"
#include "../../include/sp_prep.h"
#include "sp_common.hpp"
#include "sp_scan.hpp"
#include "sp_prep_kernels.hpp"
#include "sp_rowops.hpp"
#include "sp_sparse_kernel.hpp"
#include "sp_wave_kernel.hpp"
#include "sp_generic_kernel.hpp"
#include "sp_sddmm_kernel.hpp"

// ---------------------------------------------------------------------------------------------
// host side: C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int64_t sp_knn_workspace_bytes(const sp_knn_args *a);

namespace {

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

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

// Owned scratch of one call: a workspace the library allocated itself and the timing events.  Every early return of
// the functions below (HIP_TRY) releases them.
struct CallGuard {
    void *ws = nullptr;                  // non-null only when owned
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> events;
    int event(hipEvent_t *e) {
        *e = nullptr;
        HIP_TRY(hipEventCreate(e));
        events.push_back(*e);
        return SP_OK;
    }
    ~CallGuard() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (ws) {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(ws);
        }
    }
};

struct Config {
    int T, logT, NT, cap, hash_fill;   // generic kernel (and, unless auto-tuned apart, the sparse kernel)
    int T_s, logT_s, NT_s;             // sparse kernel: tile (region A = 8*T_s bytes) and workgroup size
    int cap_s;                     // sparse kernel's candidate buffer capacity
    int wgs_sparse, wgs_generic;   // persistent workgroups of the two row kernels
    int wgs_wave;                  // ... and of the wave-per-row kernel (single-wave workgroups), when it runs
    bool u_lds, u_lds_s;           // candidate buffer in LDS: generic / sparse kernel
    size_t lds_sparse, lds_generic;
    size_t ws_gu_bytes;     // candidate buffers in global memory for both kernels (0 when they live in LDS)
    size_t ws_gu_s_bytes;   // the sparse kernel's part of it (first)
    size_t ws_fold_bytes;   // scaled copy of m2_data when the column term is folded in, or the packed column terms (0 otherwise)
    bool pack;              // two or more column terms gathered per candidate: interleaved copy, one gather
    size_t ws_rows_bytes;   // bucket counters + work[n] + order[n] + the two descriptor queues
    size_t ws_desc_offset;  // of the sparse queue inside that block (the generic queue follows it)
    int nb_log2;            // sparse kernel: bitmap bits (log2)
    size_t ws_total;
    bool big;               // nnz(m2) >= 2^30: every row goes to the generic kernel's 64-bit-offset variant
    int n_splits;           // generic kernel: precomputed dense-window boundaries per m2 row (0 = none)
    size_t ws_split_bytes;
    int split_w;            //   fine window width (2T / f)
    int split_pmax;         // heavy generic rows are queued as up to this many pieces (ranges of fine windows), 0 = off
    int split_cap;          // at most this many rows
    size_t ws_piece_bytes;  // split_rows[cap] | piece_info[cap * pmax] | part_counts[cap * pmax] | part_cols / part_vals [cap * pmax * k]
    int items_rows;         // output slots whose work items are cut by the prepass (sp_row_items_kernel), 0 = off
    int items_stride;       // records per slot
    size_t ws_items_bytes;
    bool fold;
    bool wave;              // light rows: the wave-per-row kernel (sp_wave_kernel.hpp) runs instead of the workgroup-per-row sparse kernel
    bool duo_l;             // ... and a SECOND launch of it, in the layout with the larger collision set, takes the rows whose expected marks exceed the first's (their own queue)
    bool duo;               // the sparse kernel runs in its two-per-CU shape (512 threads, 80 KB, aliasing 2^19-bit bitmap; sp_sparse_kernel.hpp)
    size_t lds_sparse_gen;  // ... and then this is the LDS of the general variant launched beside the bounded one (the classic 512-thread layout)
    bool mono;              // the sparse kernel's monotone variant applies (val = xy / den or the raw dot, no per-row target selector)
    bool bnd;               // the sparse kernel's bounded variant is prepared and launched beside the general one (BndInfo::state picks on the device)
    size_t ws_bnd_colpack;  // offsets inside the fold block: packed id per column | packed m2 ids
    size_t ws_bnd_ids;
    bool ordered;
};

// workspace header: [0,8) queue heads sparse/generic | [8,16) queue lengths sparse/generic | [64,160) phase counters |
// [176,188) column-term minima
constexpr size_t WS_QUEUE_BYTES = 256;
constexpr size_t WS_PHASE_OFFSET = 64;
constexpr size_t WS_YMIN_OFFSET = 176;
constexpr size_t WS_FOLDZERO_OFFSET = 160;      // int: a stored entry of m2 met a zero column term while it was folded in
constexpr size_t WS_SPLITS_STATE_OFFSET = 192;  // int[2]: the dense-window boundaries exist in this workspace | workgroups of sp_m2_splits_kernel done
constexpr size_t WS_SCRATCH_OFFSET = 228;       // 28 bytes of zeroed scratch for the per-call reductions (sp_colterm_min_kernel: done | sp_bnd_xmean / range: 5 + 1 words)
constexpr size_t WS_BND_OFFSET = 200;           // BndInfo (28 bytes): the bounded variant's per-call facts, kept across SP_FLAG_REUSE_M2_PREP calls
static_assert(WS_BND_OFFSET + sizeof(BndInfo) <= WS_SCRATCH_OFFSET && WS_SCRATCH_OFFSET + 28 <= 256, "workspace header layout");
static_assert(WS_PHASE_OFFSET + PH_N * 8 <= WS_FOLDZERO_OFFSET && WS_FOLDZERO_OFFSET + 4 <= WS_YMIN_OFFSET && WS_YMIN_OFFSET + 16 <= WS_SPLITS_STATE_OFFSET &&
              WS_SPLITS_STATE_OFFSET + 8 <= WS_QUEUE_BYTES, "workspace header layout");
constexpr size_t LDS_LIMIT = 160 * 1024;

// What the library remembers about the call that BUILT the per-call passes in a caller workspace (SP_FLAG_REUSE_M2_PREP):
//   sig        a hash of everything those passes and the workspace layout depend on — m2 / Y* pointers and sizes, every scalar
//              parameter, k, the tuning fields, the flags that choose the layout.  A REUSE call with another signature is refused
//              (SP_EINVAL): it would read folded values, packed terms or window boundaries laid out for other parameters.
//   zero_term  unused since round 6 (the zero-term rerun of folding rp3beta-type calls is gone, see run_device_impl); kept for the table's layout.
// Keyed by the workspace address; an entry is rewritten by every non-REUSE call on that address, so it always describes the passes that
// are in the workspace now.  Bounded (oldest entries go first); a REUSE call on an address the table does not know is trusted as before
// (the header word at WS_FOLDZERO_OFFSET still answers the zero-term question: it is rewritten after the unfolded rerun).
struct PrepEntry { uint64_t sig; int zero_term; uint64_t seq; };
std::mutex g_prep_mu;
std::map<const void *, PrepEntry> g_prep;
uint64_t g_prep_seq = 0;
constexpr size_t PREP_TABLE_MAX = 1024;
void prep_store(const void *ws, uint64_t sig, int zero_term) {
    std::lock_guard<std::mutex> lk(g_prep_mu);
    if (g_prep.size() >= PREP_TABLE_MAX && g_prep.find(ws) == g_prep.end()) {
        auto oldest = g_prep.begin();
        for (auto it = g_prep.begin(); it != g_prep.end(); ++it) if (it->second.seq < oldest->second.seq) oldest = it;
        g_prep.erase(oldest);
    }
    g_prep[ws] = PrepEntry{sig, zero_term, ++g_prep_seq};
}
bool prep_lookup(const void *ws, PrepEntry *e) {
    std::lock_guard<std::mutex> lk(g_prep_mu);
    auto it = g_prep.find(ws);
    if (it == g_prep.end()) return false;
    *e = it->second;
    return true;
}
void prep_set_zero(const void *ws, int zero_term) {
    std::lock_guard<std::mutex> lk(g_prep_mu);
    auto it = g_prep.find(ws);
    if (it != g_prep.end()) it->second.zero_term = zero_term;
}
constexpr int ITEMS_ROWS_MAX = 1 << 21;

// LDS of the two kernels without the candidate buffer (see their carve-ups)
size_t lds_fixed_sparse(int T, int NT) { return (size_t)T * 8 + (size_t)item_cap(NT) * 16 + 4096 + CBM_BYTES + PRE_BYTES + 32 * 4 + 16 * 8; }   // (its candidate buffer lives inside region A)
size_t lds_fixed_generic(int T, int NT) { return (size_t)T * 8 + (size_t)16 * NT + 256 + 256 * 4 + 64 * 4 + 32 * 4 + 16 * 8; }

// target_cols = <matrix> as a sampled product (sp_sddmm_kernel.hpp): when the listed entries cost far less than the rows' full products.
// Sizes only (the decision must not need the device): listed entries of the targets x the average length of a column of m2, against
// MACs + the fixed toll of the row kernels.  `nnz_m2` / `n_rows_m2`: those of the call as the row kernels would see it.
bool sddmm_applies(const sp_knn_args *a, int64_t nnz_m1, int64_t nnz_m2) {
    if (a->target_col_mode != SP_SEL_MATRIX || a->k > SD_KMAX || a->n_targets <= 0 || a->n_rows_m1 <= 0 || a->n_output_cols <= 0) return false;
    if ((a->flags & (SP_FLAG_P3_PREP | SP_FLAG_NO_SPARSE_PATH)) || (a->reserved[0] & 65536)) return false;      // (bit 65536 of the ablation word: off, for A/B runs)
    const double listed = (double)a->target_col_nnz * ((double)a->n_targets / (double)a->n_rows_m1);
    const double col_len = (double)nnz_m2 / (double)a->n_output_cols;
    const double macs_row = ((double)nnz_m1 / (double)a->n_rows_m1) * ((double)nnz_m2 / (double)std::max(1, a->n_rows_m2));
    return listed * (col_len + 8.0) * 4.0 + 2000.0 * (double)a->n_targets < (double)a->n_targets * (macs_row + 30000.0);
}
// scratch of the route for an explicit m2 (its transpose + the transpose's own scratch); the flagged calls have m2^T at hand
size_t transpose_ws_bytes(long long nnz, int n_cols);
size_t sddmm_ws_bytes(const sp_knn_args *a) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return 256 + 2 * al((size_t)a->nnz_m2 * 4) + al(((size_t)a->n_output_cols + 1) * 4) + transpose_ws_bytes(a->nnz_m2, a->n_output_cols);
}

// Heavy rows of the generic kernel are queued in pieces of this many MACs (a row is cut from twice that on).  A piece is what ONE workgroup
// cannot be interrupted in: its size bounds how unevenly the persistent workgroups finish.  2^21 MACs (~1.7 ms) is nothing against the
// ~38 ms of the whole MovieLens-shaped call, and a third of an N = 8 rank's slice of it: the piece shrinks with the work a workgroup can
// expect — a quarter of it, from sizes alone —, between 2^18 and 2^21 MACs.  ONE function: the launch (sp_row_desc_kernel's split_macs) and
// the partition cost model (target_costs) must agree on which rows are cut.
unsigned split_piece_macs(const sp_knn_args *a, int wgs_generic) {
    const double avg_macs = (a->n_rows_m1 > 0 && a->n_rows_m2 > 0) ? ((double)a->nnz_m1 / a->n_rows_m1) * ((double)a->nnz_m2 / a->n_rows_m2) : 0.0;
    const double per_wg = avg_macs * (double)a->n_targets / (double)std::max(1, wgs_generic);
    unsigned piece = 1u << 21;
    while (piece > (1u << 18) && (double)piece > per_wg / 4.0) piece >>= 1;
    return piece;
}

// can the call run the sparse kernel's bounded variant (MODE 2)?  (conditions: see make_config)
bool bnd_eligible(const sp_knn_args *a, bool mono, bool fold) {
    const bool live = (a->l1 != 0.f && a->t2 != 0.f) || a->l2 != 0.f || a->l3 != 0.f;
    const bool nonneg = a->l1 >= 0.f && a->l2 >= 0.f && a->l3 >= 0.f && a->t1 >= 0.f && a->t2 >= 0.f && a->stabilized_shrink >= 0.f;
    return !mono && !fold && live && nonneg && a->a1 == 1.f && a->bayesian_shrink == 0.f && !(a->l1 * (1.f - a->t1 - a->t2) > 0.f) &&
           a->threshold >= 0.f && a->target_col_mode != SP_SEL_MATRIX &&
           a->n_output_cols > 0 && (long long)a->n_output_cols <= (1LL << BND_ID_BITS_MAX) && a->nnz_m2 > 0 &&
           !(a->flags & SP_FLAG_NO_SPARSE_PATH) && !(a->reserved[0] & 32768);      // (bit 32768 of the ablation word: off, for A/B runs)
}

int make_config(const sp_knn_args *a, int n_cus, Config *c) {
    // (threads_per_wg = 64: ask for the wave-per-row kernel wherever the call qualifies for it, whatever its average row looks like)
    const bool want_wave = a->threads_per_wg == 64;
    int NT = (a->threads_per_wg && !want_wave) ? a->threads_per_wg : 1024;   // measured best on MI355X (16 waves/CU hide the LDS/HBM round trips)
    if (NT != 256 && NT != 512 && NT != 768 && NT != 1024) return fail(SP_EINVAL, "threads_per_wg must be 64, 256, 512, 768 or 1024 (got %d)", NT);
    int T = a->table_slots ? a->table_slots : 16384;
    if (T < 1024 || (T & (T - 1))) return fail(SP_EINVAL, "table_slots must be a power of two >= 1024 (got %d)", T);
    int logT = 0;
    while ((1 << logT) < T) ++logT;
    const int load = a->load_pct > 0 ? std::min(a->load_pct, 90) : 50;

    const long long need_cap = (long long)a->k + U_SLACK;
    const size_t fixed = lds_fixed_generic(T, NT);
    if (std::max(fixed + 8 * 1024, lds_fixed_sparse(T, NT)) > LDS_LIMIT) return fail(SP_EINVAL, "table_slots=%d does not fit the 160 KiB LDS", T);
    // generic kernel's candidate buffer: LDS if k + slack entries fit beside the table, else global scratch
    long long cap_lds = (long long)((LDS_LIMIT - fixed) / 8);
    bool u_lds = need_cap <= cap_lds;
    long long cap;
    if (u_lds) {
        cap = std::max<long long>(need_cap, std::min<long long>(cap_lds, 2048));
    } else {
        cap = need_cap + 1024;
    }
    if (cap > 0x7FFFFFF0LL) return fail(SP_EINVAL, "k too large");
    // sparse kernel's candidate buffer: the last quarter of region A when SEL_E*NT entries (what its register-resident
    // selection handles) fit there and leave room above k; else global scratch
    // The sparse kernel's own shape.  Its column bitmap wants one bit per output column: up to 2^18 columns fit a
    // 32 KiB region A, and then THREE 256-thread workgroups share a CU (53.5 KB of LDS each) instead of one of 1024
    // threads — the dense phases of one overlap with the sweeps of the others, and a 4-wave barrier is cheap
    // (user-scoring slice, 100k items: 97 -> 50 ms).  Needs k + 512 <= 1024 for the candidate buffer to stay in LDS.
    int NT_s = NT, T_s = T, logT_s = logT;
    // ... and for the typical row to stay on this kernel with the smaller collision set (rows are classified one by one
    // on the device: expected colliding products MACs^2 / (2 n_cols) <= 0.3 * slots; here the average row, from sizes alone)
    const double avg_macs = (a->n_rows_m1 > 0 && a->n_rows_m2 > 0) ? ((double)a->nnz_m1 / a->n_rows_m1) * ((double)a->nnz_m2 / a->n_rows_m2) : 0.0;
    // (beyond 2^18 columns the bitmap aliases — columns modulo its size — which only adds expected collisions)
    const bool small_rows = avg_macs * avg_macs / (2.0 * std::max(1, std::min(a->n_output_cols, 1 << 18))) <= 0.25 * 1024.0;
    if ((!a->threads_per_wg || want_wave) && !a->table_slots && (long long)a->k + 512 <= (long long)SEL_E * 256 && small_rows) {
        NT_s = 256; T_s = 4096; logT_s = 12;
    }
    // Rows of the headline's weight (C2: 41 k products over 10^6 columns) are too heavy for that shape and ran ONE 1024-thread workgroup per
    // CU (128 KB exact bitmap).  Round 6: TWO 512-thread workgroups per CU with a 2^19-bit aliasing bitmap (DUO, sp_sparse_kernel.hpp) when
    // the variant is of the monotone type (decided below), k leaves room in its 2048-entry candidate buffer and the AVERAGE row's expected
    // marked columns  MACs^2 / (2 * bitmap bits)  fit its 2048 rank-addressed slots with room to spare (rows are classified one by one on the device).
    // (bit 524288 of the ablation word: off, for A/B runs)
    const bool any_norm0 = a->l1 != 0.f || a->l2 != 0.f || a->l3 != 0.f || a->stabilized_shrink != 0.f || a->bayesian_shrink != 0.f;
    const bool fold0 = !(a->flags & SP_FLAG_NO_FOLD) && a->l1 == 0.f && a->a1 == 1.f && a->stabilized_shrink == 0.f &&
                       a->bayesian_shrink == 0.f && ((a->l2 != 0.f) != (a->l3 != 0.f)) && a->nnz_m2 > 0;
    const bool mono0 = (fold0 || !any_norm0) && a->target_col_mode != SP_SEL_MATRIX;
    bool duo = false;
    int duo_direct = DUO_CS_DIRECT;
    const bool big0 = a->nnz_m2 >= (1LL << 30) - 1024 || (a->reserved[0] & 1024);      // (no sparse kernel runs at all, see below)
    if ((mono0 || bnd_eligible(a, mono0, fold0)) && !big0 && !(a->flags & SP_FLAG_NO_SPARSE_PATH) &&
        !a->threads_per_wg && !a->table_slots && NT_s == 1024 && !(a->reserved[0] & 524288) && (long long)a->k + 512 <= (long long)(DUO_U_BYTES / 8) &&
        a->n_output_cols > (1 << 16) && avg_macs > 0.0) {
        const double bits = (double)std::min<long long>(a->n_output_cols, 1LL << DUO_NB_LOG2);
        const double marks = avg_macs * avg_macs / (2.0 * bits);
        duo = marks <= 0.82 * (double)DUO_CS_DIRECT || (marks <= 0.82 * (double)DUO_CS_DIRECT_L && (long long)a->k + 512 <= (long long)DUO_U_ENTRIES_L);
        // (T_s of this shape = the rank-addressed slots of its collision set: 2048, or — between 1.7 k and 2.5 k expected marks per row, where
        // every row used to go to the generic kernel: 109 ms against 21.6 per 200 k rows of 41 k products over 400 k columns — 3072 with 1024
        // overflow slots, a member pool of 2048 entries instead of 3072 and 1536 entries of U instead of 2048)
        duo_direct = marks <= 0.82 * (double)DUO_CS_DIRECT ? DUO_CS_DIRECT : DUO_CS_DIRECT_L;
    }
    if (duo) { NT_s = DUO_NT; T_s = duo_direct; logT_s = 13; }      // (logT_s = 13: the 2^19-bit bitmap; the kernel's own logT is set where its parameters are filled)
    const bool u_lds_s = duo || (((size_t)SEL_E * NT_s * 8 <= (size_t)T_s * 2) && ((long long)a->k + 512 <= (long long)SEL_E * NT_s));
    const long long cap_s = duo ? (long long)(duo_direct == DUO_CS_DIRECT ? DUO_U_BYTES / 8 : DUO_U_ENTRIES_L) : u_lds_s ? (long long)SEL_E * NT_s : ((need_cap + 1024) & ~1LL);
    c->T = T; c->logT = logT; c->NT = NT; c->cap = (int)cap; c->u_lds = u_lds; c->cap_s = (int)cap_s; c->u_lds_s = u_lds_s;
    c->T_s = T_s; c->logT_s = logT_s; c->NT_s = NT_s;
    c->hash_fill = std::max(1, (int)((long long)T * load / 100));
    c->lds_sparse = duo ? sp_duo_lds_bytes() : lds_fixed_sparse(T_s, NT_s);
    c->lds_sparse_gen = lds_fixed_sparse(duo ? 8192 : T_s, NT_s);
    c->duo = duo;
    // Rows are classified one by one: a call whose AVERAGE row fits the 2048 rank-addressed slots still has rows that do not (real data has
    // row degrees: a binary matrix with Poisson(64) rows sent a quarter of them — 43 k to 53 k products — to the generic kernel, 32 of the
    // call's 54 ms).  Those rows get a queue of their own (the wave kernel's: it never runs beside this shape) and a second launch of the
    // same kernel in the larger layout.
    c->duo_l = duo && duo_direct == DUO_CS_DIRECT && (long long)a->k + 512 <= (long long)DUO_U_ENTRIES_L && !(a->reserved[0] & 1048576);      // (bit 1048576 of the ablation word: off)
    c->lds_generic = lds_fixed_generic(T, NT) + (u_lds ? (size_t)cap * 8 : 0);
    auto wgs_for = [&](size_t lds, int nt) {
        int per_cu = (int)std::max<size_t>(1, LDS_LIMIT / lds);
        per_cu = std::min(per_cu, 2048 / nt);
        per_cu = std::max(1, std::min(per_cu, 8));
        int n = a->num_wgs > 0 ? a->num_wgs : n_cus * per_cu;
        return std::max(1, std::min(n, std::max(1, a->n_targets)));
    };
    c->wgs_sparse = wgs_for(c->lds_sparse, NT_s);
    c->wgs_generic = wgs_for(c->lds_generic, NT);
    c->ws_gu_s_bytes = u_lds_s ? 0 : (((size_t)c->wgs_sparse * (size_t)cap_s * 8 + 255) & ~(size_t)255);
    c->ws_gu_bytes = c->ws_gu_s_bytes + (u_lds ? 0 : (((size_t)c->wgs_generic * (size_t)cap * 8 + 255) & ~(size_t)255));
    // product-form epilogue  val = xy / (l * X[t] * Y[c])  (cosine, asymmetric cosine, rp3beta without shrink):
    // Y is divided into the m2 values once per call, the kernels then need no column-term gathers at all
    c->fold = !(a->flags & SP_FLAG_NO_FOLD) && a->l1 == 0.f && a->a1 == 1.f && a->stabilized_shrink == 0.f &&
              a->bayesian_shrink == 0.f && ((a->l2 != 0.f) != (a->l3 != 0.f)) && a->nnz_m2 > 0;
    const bool any_norm = a->l1 != 0.f || a->l2 != 0.f || a->l3 != 0.f || a->stabilized_shrink != 0.f || a->bayesian_shrink != 0.f;
    c->mono = (c->fold || !any_norm) && a->target_col_mode != SP_SEL_MATRIX;      // (a MATRIX filter is handled through the collision bitmap)
    c->pack = !c->fold && ((a->l1 != 0.f) + (a->l2 != 0.f) + (a->l3 != 0.f) >= 2) && a->n_output_cols > 0;
    c->ws_fold_bytes = c->fold ? (((size_t)a->nnz_m2 * 4 + 255) & ~(size_t)255) : c->pack ? (((size_t)a->n_output_cols * 16 + 255) & ~(size_t)255) : 0;
    // Bounded variant of the sparse kernel (MODE 2): a general epilogue whose value is bounded through ONE per-column term carried in the
    // upper 12 bits of the m2 column ids.  Needs: column terms that are live and not folded, non-negative weights (the bound), a1 = 1, no
    // Bayesian factor, a denominator that does not grow with the raw dot (t1 + t2 >= 1 whenever l1 != 0), threshold >= 0 (negative values
    // are never wanted), no per-row TARGET matrix (a MATRIX filter goes through the collision bitmap, as in the monotone variant), ids of at most 22 bits (the code keeps 12 / 11 / 10 bits).  What it cannot serve runs on the general variant.
    {
        c->bnd = bnd_eligible(a, c->mono, c->fold);
        c->ws_bnd_colpack = c->ws_bnd_ids = 0;
        if (c->bnd) {
            c->ws_bnd_colpack = c->ws_fold_bytes;
            c->ws_bnd_ids = c->ws_bnd_colpack + (((size_t)a->n_output_cols * 4 + 255) & ~(size_t)255);
            c->ws_fold_bytes = c->ws_bnd_ids + (((size_t)a->nnz_m2 * 4 + 255) & ~(size_t)255);
        }
    }
    c->ordered = !(a->flags & (SP_FLAG_STATIC_SCHED | SP_FLAG_NO_ROW_ORDER)) && a->n_targets > std::min(c->wgs_sparse, c->wgs_generic);
    // 512 B of bucket counters | work[n] | order[n] | (32-byte aligned) sparse queue n x 32 B | wave queue n x 32 B | generic queue n x 32 B
    c->ws_desc_offset = (512 + (size_t)a->n_targets * 8 + 31) & ~(size_t)31;
    c->ws_rows_bytes = (c->ws_desc_offset + (size_t)a->n_targets * 96 + 255) & ~(size_t)255;
    // sparse kernel: one bit per column while the columns fit region A, else columns alias modulo the bitmap size
    int nb = 10;
    while (nb < c->logT_s + 6 && (1LL << nb) < (long long)a->n_output_cols) ++nb;
    c->nb_log2 = nb;
    // generic kernel, standard dense windows of 2T columns: their boundaries inside every m2 row, found once per call — at a
    // finer grain (2T / f) when that stays a short list, so that heavy rows can be cut into pieces narrower than a window
    {
        const long long Td = 2LL * T;
        c->n_splits = 0; c->split_w = (int)Td;
        if ((long long)a->n_output_cols > Td && a->n_rows_m2 > 0 && a->nnz_m2 > 0) {
            for (int f = 8; f >= 1; f >>= 1) {      // (f = 8 since round 5: the heaviest item of the MovieLens shape in 21 pieces instead of 11 — a piece is the unit the workgroups balance with)
                const long long G = Td / f, nsp = ((long long)a->n_output_cols + G - 1) / G - 1;
                if (nsp >= 1 && nsp <= 31) { c->n_splits = (int)nsp; c->split_w = (int)G; break; }
            }
        }
        c->ws_split_bytes = c->n_splits ? (((size_t)a->n_rows_m2 * (size_t)c->n_splits * 4 + 255) & ~(size_t)255) : 0;
    }
    // heavy generic rows (a popular item of a ratings matrix: one row can be a third of the kernel's time on one workgroup) are
    // queued as one piece per standard dense window; needs the per-call boundaries above and a merge buffer of pieces * k records
    c->split_pmax = 0; c->split_cap = 0; c->ws_piece_bytes = 0;
    {
        const int pmax = (int)std::min<long long>(c->n_splits + 1, 8192 / std::max(1, a->k));
        if (c->n_splits >= 1 && pmax >= 2 && !(a->reserved[0] & 4096)) {      // (bit 4096 of the ablation word: off)
            c->split_pmax = pmax;
            c->split_cap = std::min(a->n_targets, 2048);
            const size_t np = (size_t)c->split_cap * (size_t)c->split_pmax;
            c->ws_piece_bytes = (((size_t)c->split_cap * 16 + np * 8 + np * 4 + np * (size_t)a->k * 8) + 255) & ~(size_t)255;
            c->ws_rows_bytes += (np * 32 + 255) & ~(size_t)255;       // room for the extra entries of the generic queue (the last array of that block)
        }
    }
    // nnz(m2) >= 2^30: the sparse kernel's 32-bit buffer offsets do not reach; every row takes the generic kernel's 64-bit-offset
    // variant.  (This assignment was lost in round 2's piece splitter commit: `big` was stack garbage from then on — the tests that
    // need it passed by the accident of what the stack held; round 3's cache cap changed that accident and exposed it.)
    c->big = a->nnz_m2 >= (1LL << 30) - 1024 || (a->reserved[0] & 1024);      // (bit 1024 of the ablation word: force it, for tests at small sizes)
    if (c->big && c->bnd) {      // (no sparse kernel runs at all: nothing to prepare)
        c->bnd = false;
        c->ws_fold_bytes = c->ws_bnd_colpack;
    }
    // the sparse kernel's work items, cut once per call: ITEMS_STRIDE * 16 B = 4 KB per output slot, for at most ITEMS_ROWS_MAX slots (the rows beyond
    // are set up in the kernel, as are rows of more than 64 entries or more than ITEMS_PRE items)
    c->items_rows = (!(a->flags & SP_FLAG_NO_SPARSE_PATH) && !c->big && !(a->reserved[0] & 2048) && a->nnz_m2 > 0) ? std::min(a->n_targets, ITEMS_ROWS_MAX) : 0;
    // Records per slot by need (round 5; VERDICT r4 #9: 4 KB per slot whatever the rows hold): the average row's records from sizes — one
    // trip per 256 elements of a segment, or, where trips are packed (the 256-thread shape, segments shorter than a trip), a trip per 64
    // lanes of the virtual lane axis and up to one second-piece record each — x 1.5, in a stride of 64 / 128 / 256 records.  A row that
    // needs more than its slot holds is set up in the kernel, as rows beyond ITEMS_PRE records always were.
    {
        const double n1 = std::min(64.0, a->n_rows_m1 > 0 ? (double)a->nnz_m1 / a->n_rows_m1 : 0.0);
        const double len2 = a->n_rows_m2 > 0 ? (double)a->nnz_m2 / a->n_rows_m2 : 0.0;
        const double trips_u = n1 * std::max(1.0, std::ceil(len2 / 256.0));
        const double trips_p = std::ceil(n1 * std::ceil(len2 / 4.0) / 64.0) + 2.0;
        const bool packs = NT_s == 256 && 4.0 * trips_p <= 3.0 * trips_u;
        const double need = 1.5 * (packs ? 2.0 * trips_p + 2.0 : trips_u + 2.0);
        c->items_stride = need <= 63.0 ? 64 : need <= 127.0 ? 128 : ITEMS_STRIDE;
    }
    // Light rows (user scoring: a few thousand products, k <= 128, monotone epilogue): one WAVE per row, nine to twelve rows in flight per
    // CU (sp_wave_kernel.hpp) — when the average row fits its 63 packed trips with room to spare, or on request.  Up to 2^17 output columns
    // the wave's column bitmap is exact; beyond, columns alias modulo 2^17 (an aliased column only takes the collision-set route, where
    // sums are kept per column: exact) and sp_row_desc_kernel sends the kernel the rows whose expected marks fit its collision set.
    c->wave = c->items_rows > 0 && c->mono && NT_s == 256 && a->n_output_cols > T && a->k <= WV_KMAX &&
              !(a->reserved[0] & 16384) && (want_wave || (!a->threads_per_wg && avg_macs <= 10000.0));
    c->wgs_wave = 0;
    // (a wave call's records are one per SEGMENT, 64 x 12 bytes per row — sp_row_items_wave_kernel; the few rows its workgroup-per-row
    // companion takes need more than such a slot holds and are set up in the kernel)
    if (c->wave) c->items_stride = WAVE_ITEMS_STRIDE;
    c->ws_items_bytes = ((size_t)c->items_rows * (size_t)c->items_stride * 16 + 255) & ~(size_t)255;
    if (c->wave) {
        // (the workgroup-per-row kernel keeps its 256-thread shape beside it: sparse rows the wave kernel does not take — more than 64 m1
        // entries, more products than its 63 trips hold — have a queue of their own and run there, as in round 3)
        const int wv_a = wv_region_bytes(a->n_output_cols);      // the column bitmap: twelve, eleven, ten or nine rows in flight per CU
        c->wgs_wave = std::max(1, std::min(a->num_wgs > 0 ? a->num_wgs : n_cus * (int)(LDS_LIMIT / wv_lds_bytes(wv_a)), std::max(1, a->n_targets)));
    }
    c->ws_total = WS_QUEUE_BYTES + c->ws_gu_bytes + c->ws_fold_bytes + c->ws_rows_bytes + c->ws_split_bytes + c->ws_piece_bytes + c->ws_items_bytes;
    if (sddmm_applies(a, a->nnz_m1, a->nnz_m2)) c->ws_total = std::max(c->ws_total, sddmm_ws_bytes(a));      // (explicit m2: the route transposes it)
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
    const bool m2t = (a->flags & SP_FLAG_M2_IS_M1_T) != 0;     // m2 = m1^T, built on the device: the m2_* pointers and nnz_m2 are ignored
    const bool m1t = (a->flags & SP_FLAG_M1_IS_M2_T) != 0;     // m1 = m2^T, built on the device: the m1_* pointers and nnz_m1 are ignored
    const bool dev_norms = (a->flags & SP_FLAG_NORMS_ON_DEVICE) != 0;
    if (m2t && m1t) return fail(SP_EINVAL, "SP_FLAG_M2_IS_M1_T and SP_FLAG_M1_IS_M2_T exclude each other");
    if ((a->flags & (SP_FLAG_P3_PREP | SP_FLAG_DEPOP_ROWSUM)) && !m2t && !m1t)
        return fail(SP_EINVAL, "SP_FLAG_P3_PREP / SP_FLAG_DEPOP_ROWSUM need SP_FLAG_M2_IS_M1_T or SP_FLAG_M1_IS_M2_T");
    if ((a->flags & SP_FLAG_NORMS_ON_DEVICE) && !m2t && !m1t && a->on_device)
        return fail(SP_EINVAL, "SP_FLAG_NORMS_ON_DEVICE with an explicit m2 is a host-mode option (device mode: SP_FLAG_M2_IS_M1_T or SP_FLAG_M1_IS_M2_T)");
    if ((a->flags & SP_FLAG_DEPOP_ROWSUM) && !(a->flags & SP_FLAG_P3_PREP))
        return fail(SP_EINVAL, "SP_FLAG_DEPOP_ROWSUM needs SP_FLAG_P3_PREP");
    if (a->col_keep && !m2t && (a->on_device || m1t))
        return fail(SP_EINVAL, "col_keep with an explicit m2 is a host-mode option (device-resident m2 is filtered by its owner)");
    if (a->col_keep && (a->flags & SP_FLAG_P3_PREP) && !m2t)
        return fail(SP_EINVAL, "col_keep with SP_FLAG_P3_PREP needs SP_FLAG_M2_IS_M1_T (the columns are dropped from the m2 built here, after its rows were normalised)");
    if ((a->flags & (SP_FLAG_CSR_OUT | SP_FLAG_CHECK_ZEROS | SP_FLAG_BINARY | SP_FLAG_CHECK_SORTED)) && a->on_device)
        return fail(SP_EINVAL, "SP_FLAG_CSR_OUT / SP_FLAG_CHECK_ZEROS / SP_FLAG_BINARY / SP_FLAG_CHECK_SORTED are host-mode flags (on_device = 0)");
    if ((a->flags & SP_FLAG_CSR_OUT) && a->n_targets > 0 && !a->csr_indptr) return fail(SP_EINVAL, "SP_FLAG_CSR_OUT needs csr_indptr");
    if ((m2t || m1t) && a->n_output_cols != a->n_rows_m1)
        return fail(SP_EINVAL, "SP_FLAG_M2_IS_M1_T / SP_FLAG_M1_IS_M2_T: n_output_cols (%d) must equal n_rows_m1 (%d)", a->n_output_cols, a->n_rows_m1);
    if (a->n_targets > 0) {
        if (!a->targets || (!m1t && !a->m1_indptr) || (!m2t && !a->m2_indptr) || !a->cols || !a->values)
            return fail(SP_EINVAL, "NULL input/output pointer");
        if (!a->rows && !(a->flags & (SP_FLAG_NO_ROWS_OUT | SP_FLAG_CSR_OUT)))
            return fail(SP_EINVAL, "rows is NULL");
        if (!m1t && a->nnz_m1 > 0 && (!a->m1_data || !a->m1_indices)) return fail(SP_EINVAL, "m1 arrays NULL");
        if (!m2t && a->nnz_m2 > 0 && (!a->m2_data || !a->m2_indices)) return fail(SP_EINVAL, "m2 arrays NULL");
        if (!dev_norms && a->l1 != 0.f && (!a->Xtversky || !a->Ytversky)) return fail(SP_EINVAL, "l1 != 0 needs Xtversky/Ytversky");
        if (!dev_norms && a->l2 != 0.f && (!a->Xcosine || !a->Ycosine)) return fail(SP_EINVAL, "l2 != 0 needs Xcosine/Ycosine");
        if (a->l3 != 0.f && (!a->Xdepop || (!a->Ydepop && !(a->flags & SP_FLAG_DEPOP_ROWSUM)))) return fail(SP_EINVAL, "l3 != 0 needs Xdepop/Ydepop");
        if (a->filter_mode == SP_SEL_MATRIX && (!a->filter_m_indptr || (a->filter_nnz > 0 && !a->filter_m_indices)))
            return fail(SP_EINVAL, "filter MATRIX mode needs indptr/indices");
        if (a->target_col_mode == SP_SEL_MATRIX && (!a->target_col_m_indptr || (a->target_col_nnz > 0 && !a->target_col_m_indices)))
            return fail(SP_EINVAL, "target MATRIX mode needs indptr/indices");
    }
    if (a->filter_mode < 0 || a->filter_mode > 2 || a->target_col_mode < 0 || a->target_col_mode > 2)
        return fail(SP_EINVAL, "bad selector mode");
    if (a->n_devices < 0 || a->n_devices > 64) return fail(SP_EINVAL, "n_devices must be in [0, 64] (got %d)", a->n_devices);
    return SP_OK;
}

int device_cus(int device, int *n_cus) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    *n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // SIMILARIPY_AMD_RESERVE_CUS=n: the persistent row kernels are sized for n CUs fewer.  They fill every CU they are given (LDS), and a
    // kernel of another stream — the RCCL gather of a finished sub-slab in the multi-GPU step — only starts when workgroups retire:
    // a few CUs left free are what lets the communication actually run beside the next sub-launch (distributed.py, bench.py --gpus N).
    if (const char *e = getenv("SIMILARIPY_AMD_RESERVE_CUS")) {
        const int r = atoi(e);
        if (r > 0) *n_cus = std::max(1, *n_cus - r);
    }
    return SP_OK;
}

template <int NT>
int launch_sparse(const KParams &kp, const Config &c, hipStream_t stream) {
    if constexpr (NT == DUO_NT) {
        if (c.duo) {
            // the two-per-CU shape (monotone or bounded variant; the general variant that backs the bounded one up — BndInfo::state != 1: a
            // zero or negative column term, rare — runs the classic 512-thread layout on the same parameters, one workgroup per CU)
            auto one = [&](const KParams &kq, bool second) -> int {
                auto kd = second ? (c.bnd ? sp_knn_sparse_kernel<DUO_NT, true, 2, true, true> : sp_knn_sparse_kernel<DUO_NT, true, 1, true, true>)
                                 : (c.bnd ? sp_knn_sparse_kernel<DUO_NT, true, 2, true> : sp_knn_sparse_kernel<DUO_NT, true, 1, true>);
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_sparse));
                hipLaunchKernelGGL(kd, dim3(c.wgs_sparse), dim3(NT), c.lds_sparse, stream, kq);
                HIP_TRY(hipGetLastError());
                if (c.bnd) {
                    auto kg = sp_knn_sparse_kernel<DUO_NT, true, 0>;
                    KParams kpg = kq;      // (the classic layout reads its region size from T: 64 KB = the 2^19-bit bitmap; the DUO kernel keeps its slot count there)
                    kpg.T = 8192; kpg.logT = 13;
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_sparse_gen));
                    hipLaunchKernelGGL(kg, dim3(std::max(1, c.wgs_sparse / 2)), dim3(NT), c.lds_sparse_gen, stream, kpg);
                    HIP_TRY(hipGetLastError());
                }
                return SP_OK;
            };
            TRY(one(kp, false));
            if (c.duo_l) {
                // the rows whose expected marks need the larger collision set: their own queue (head, length, descriptors: the wave kernel's
                // words of the workspace), the same kernel in its other layout — 3072 + 1024 slots, a 2048-entry pool, 1536 entries of U
                KParams kl = kp;
                kl.T = DUO_CS_DIRECT_L; kl.logT = 10; kl.cap_s = DUO_U_ENTRIES_L;
                kl.queue = kp.queue + 6;
                kl.qcount = kp.queue + 7;
                kl.desc = kp.desc + 2 * (size_t)kp.n_targets;
                TRY(one(kl, true));
            }
            return SP_OK;
        }
    }
    if (c.bnd) {
        // the bounded variant; BndInfo::state (written by the per-call passes on the device) decides at its first instruction whether it
        // or the general variant launched right behind it does the rows — no read-back, no synchronisation
        auto kb = c.u_lds_s ? sp_knn_sparse_kernel<NT, true, 2> : sp_knn_sparse_kernel<NT, false, 2>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_sparse));
        hipLaunchKernelGGL(kb, dim3(c.wgs_sparse), dim3(NT), c.lds_sparse, stream, kp);
        HIP_TRY(hipGetLastError());
    }
    auto ks = c.mono ? (c.u_lds_s ? sp_knn_sparse_kernel<NT, true, 1> : sp_knn_sparse_kernel<NT, false, 1>)
                     : (c.u_lds_s ? sp_knn_sparse_kernel<NT, true, 0> : sp_knn_sparse_kernel<NT, false, 0>);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_sparse));
    hipLaunchKernelGGL(ks, dim3(c.wgs_sparse), dim3(NT), c.lds_sparse, stream, kp);
    HIP_TRY(hipGetLastError());
    return SP_OK;
}

template <int NT>
int launch_generic(const KParams &kp, const Config &c, hipStream_t stream) {
    auto kg = c.big ? (c.u_lds ? sp_knn_generic_kernel<NT, true, true> : sp_knn_generic_kernel<NT, false, true>)
                    : (c.u_lds ? sp_knn_generic_kernel<NT, true> : sp_knn_generic_kernel<NT, false>);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds_generic));
    hipLaunchKernelGGL(kg, dim3(c.wgs_generic), dim3(NT), c.lds_generic, stream, kp);
    HIP_TRY(hipGetLastError());
    return SP_OK;
}

// the per-call pass the generic kernel alone needs (sp_m2_splits_kernel), queued between the sparse-row kernels and the generic one
struct SplitsLaunch { int n_rows_m2; const int *m2_indptr, *m2_indices; int split_w, n_splits; int *out; const unsigned *qcount_g; int *state; };

// kp_s: the sparse kernel's parameters (its own tile), kp: the generic kernel's
int launch_rows(const KParams &kp_s, const KParams &kp, const Config &c, hipStream_t stream, hipEvent_t *ev /* [4] or NULL: around the two row kernels */,
                const SplitsLaunch *sl = nullptr) {
    // sparse rows first; what it cannot finish joins the generic queue, which the second launch drains
    if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
    if (kp.sparse_path && c.wave) {
        KParams kp_w = kp_s;                       // its own queue: head, length, descriptors
        kp_w.queue = kp_s.queue + 6;
        kp_w.qcount = kp_s.queue + 7;
        kp_w.desc = kp_s.desc + 2 * (size_t)kp_s.n_targets;
        if (wv_region_bytes(kp_s.n_cols) == WV_A_TIGHT) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_knn_wave_kernel<WV_A_TIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, wv_lds_bytes(WV_A_TIGHT)));
            hipLaunchKernelGGL(sp_knn_wave_kernel<WV_A_TIGHT>, dim3(c.wgs_wave), dim3(64), wv_lds_bytes(WV_A_TIGHT), stream, kp_w);
        } else if (wv_region_bytes(kp_s.n_cols) == WV_A_SMALL) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_knn_wave_kernel<WV_A_SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, wv_lds_bytes(WV_A_SMALL)));
            hipLaunchKernelGGL(sp_knn_wave_kernel<WV_A_SMALL>, dim3(c.wgs_wave), dim3(64), wv_lds_bytes(WV_A_SMALL), stream, kp_w);
        } else if (wv_region_bytes(kp_s.n_cols) == WV_A_MID) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_knn_wave_kernel<WV_A_MID>), hipFuncAttributeMaxDynamicSharedMemorySize, wv_lds_bytes(WV_A_MID)));
            hipLaunchKernelGGL(sp_knn_wave_kernel<WV_A_MID>, dim3(c.wgs_wave), dim3(64), wv_lds_bytes(WV_A_MID), stream, kp_w);
        } else {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_knn_wave_kernel<WV_A_LARGE>), hipFuncAttributeMaxDynamicSharedMemorySize, wv_lds_bytes(WV_A_LARGE)));
            hipLaunchKernelGGL(sp_knn_wave_kernel<WV_A_LARGE>, dim3(c.wgs_wave), dim3(64), wv_lds_bytes(WV_A_LARGE), stream, kp_w);
        }
        HIP_TRY(hipGetLastError());
    }
    if (kp.sparse_path) {
        int rc;
        if (c.NT_s == 256) rc = launch_sparse<256>(kp_s, c, stream);
        else if (c.NT_s == 512) rc = launch_sparse<512>(kp_s, c, stream);
        else if (c.NT_s == 768) rc = launch_sparse<768>(kp_s, c, stream);
        else rc = launch_sparse<1024>(kp_s, c, stream);
        if (rc) return rc;
    }
    if (ev) HIP_TRY(hipEventRecord(ev[1], stream));
    if (sl) {
        hipLaunchKernelGGL(sp_m2_splits_kernel, dim3((unsigned)std::max(1, std::min(256 * 16, (sl->n_rows_m2 + 3) / 4))), dim3(256), 0, stream, sl->n_rows_m2, sl->m2_indptr,
                           sl->m2_indices, sl->split_w, sl->n_splits, sl->out, sl->qcount_g, sl->state);
        HIP_TRY(hipGetLastError());
    }
    if (ev) HIP_TRY(hipEventRecord(ev[2], stream));
    int rc;
    if (c.NT == 256) rc = launch_generic<256>(kp, c, stream);
    else if (c.NT == 512) rc = launch_generic<512>(kp, c, stream);
    else if (c.NT == 768) rc = launch_generic<768>(kp, c, stream);
    else rc = launch_generic<1024>(kp, c, stream);
    if (rc) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[3], stream));
    return SP_OK;
}

// A host-mode call may cut its target list into chunks (sub-launches that reuse the first one's passes over m2, SP_FLAG_REUSE_M2_PREP)
// so that a chunk's results travel to the host while the next chunk computes: after_launch(j) is called when chunk j's launches are
// queued (it records an event on the stream).
struct ChunkHook {
    int n_chunks = 1;
    std::vector<size_t> bounds;                     // [n_chunks + 1] slots
    std::function<int(int)> after_launch;
};

// Everything the per-call passes over m2 / Y* and the layout of the workspace blocks in front of the per-target state depend on (FNV-1a).
uint64_t prep_signature(const sp_knn_args *a, const Config &c) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
#define SP_MIX(x) mix(&(x), sizeof(x))
    const uint32_t fl = a->flags & (SP_FLAG_NO_FOLD | SP_FLAG_NO_SPARSE_PATH);
    const int64_t abl = a->reserved[0] & (1024 | 2048 | 4096 | 16384 | 32768 | 65536 | 524288 | 1048576);      // (items_stride follows from sizes the signature covers)
    SP_MIX(fl); SP_MIX(abl);
    SP_MIX(a->n_rows_m2); SP_MIX(a->n_output_cols); SP_MIX(a->nnz_m2);
    SP_MIX(a->m2_data); SP_MIX(a->m2_indices); SP_MIX(a->m2_indptr);
    SP_MIX(a->Ytversky); SP_MIX(a->Ycosine); SP_MIX(a->Ydepop);
    SP_MIX(a->a1); SP_MIX(a->l1); SP_MIX(a->l2); SP_MIX(a->l3); SP_MIX(a->t1); SP_MIX(a->t2);
    SP_MIX(a->stabilized_shrink); SP_MIX(a->bayesian_shrink);
    SP_MIX(a->k); SP_MIX(a->table_slots); SP_MIX(a->threads_per_wg); SP_MIX(a->load_pct);
    const uint64_t lay[5] = {(uint64_t)c.ws_fold_bytes, (uint64_t)c.ws_split_bytes, (uint64_t)c.n_splits, (uint64_t)c.split_w, (uint64_t)(c.fold ? 1 : 0) | (c.pack ? 2 : 0) | (c.bnd ? 4 : 0)};
    mix(lay, sizeof(lay));
#undef SP_MIX
    return h;
}

// all pointers in `a` are device pointers here
// (sig_override: the unfolded rerun of a folding call keeps the signature of the call as the caller made it)
int run_device_impl(sp_knn_args *a, const uint64_t *sig_override = nullptr) {
    HIP_TRY(hipSetDevice(a->device));
    if (a->n_targets == 0) { a->kernel_ms = 0.f; return SP_OK; }
    int n_cus = 256;
    int rc = device_cus(a->device, &n_cus);
    if (rc) return rc;
    Config c{};
    rc = make_config(a, n_cus, &c);
    if (rc) return rc;

    hipStream_t stream = (hipStream_t)a->stream;
    unsigned char *ws = (unsigned char *)a->workspace;
    CallGuard guard;
    guard.stream = stream;
    if (!ws) {
        hipError_t me = hipMalloc((void **)&ws, c.ws_total);
        if (me != hipSuccess) {      // out of memory: the host-mode buffer cache may be what holds it
            (void)hipGetLastError();
            (void)sp_device_cache_trim();
            me = hipMalloc((void **)&ws, c.ws_total);
        }
        if (me != hipSuccess) return fail(SP_ENOMEM, "hipMalloc(%zu bytes of workspace) failed: %s", c.ws_total, hipGetErrorString(me));
        guard.ws = ws;
    } else if (a->workspace_bytes < (int64_t)c.ws_total) {
        return fail(SP_EWORKSPACE, "workspace too small: need %zu bytes, got %lld", c.ws_total, (long long)a->workspace_bytes);
    }

    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) {
        TRY(guard.event(&ev0));
        TRY(guard.event(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }

    // SP_FLAG_REUSE_M2_PREP: an earlier call on this workspace left the per-call passes over m2 / Y* behind (folded values or packed
    // column terms, their minima, the dense-window boundaries, the sign flag); only the per-target state is rebuilt
    const bool reuse = (a->flags & SP_FLAG_REUSE_M2_PREP) != 0 && a->workspace != nullptr;
    const uint64_t sig = sig_override ? *sig_override : prep_signature(a, c);
    PrepEntry built{};
    const bool known = a->workspace != nullptr && prep_lookup(ws, &built);
    if (reuse && known && built.sig != sig)
        return fail(SP_EINVAL, "SP_FLAG_REUSE_M2_PREP: m2 / Y*, a scalar parameter, k or a tuning field differs from the call that built the passes "
                               "in this workspace — drop the flag (the passes are rebuilt) or repeat that call's arguments");
    if (a->workspace && !reuse) prep_store(ws, sig, -1);
    HIP_TRY(hipMemsetAsync(ws, 0, reuse ? WS_FOLDZERO_OFFSET : WS_QUEUE_BYTES, stream));
    // header | blocks that depend on m2 and the parameters only (same offsets whatever the target list) | blocks sized by n_targets
    unsigned char *ws_fold = ws + WS_QUEUE_BYTES;
    int *ws_split = (int *)(ws_fold + c.ws_fold_bytes);
    unsigned char *ws_gu = (unsigned char *)ws_split + c.ws_split_bytes;
    unsigned char *ws_rows = ws_gu + c.ws_gu_bytes;
    unsigned char *ws_piece = ws_rows + c.ws_rows_bytes;
    unsigned char *ws_items = ws_piece + c.ws_piece_bytes;

    // minima of the column-term vectors feed the gather-free upper bound (Epi::upper); it is sound only
    // when every weight / shrink is non-negative (NaN parameters fail the comparisons and disable it)
    const bool bound_ok = (a->l1 >= 0.f) && (a->l2 >= 0.f) && (a->l3 >= 0.f) && (a->t1 >= 0.f) && (a->t2 >= 0.f) &&
                          (a->stabilized_shrink >= 0.f) && (a->bayesian_shrink >= 0.f);
    float *ymin_dev = (float *)(ws + WS_YMIN_OFFSET);
    float *folded = nullptr;
    float4 *ypack = nullptr;
    if (c.fold) {
        folded = (float *)ws_fold;
        // (a depopularisation weight can be exactly 0 on a column that has entries — a 'sum' weight of signed data.  The reference then
        // reports value 0 for every such column a product touches (zero denominator -> 0, s_plus.h:144-150); the fold writes 0.0 for the
        // entries of such a column, so every product on it is 0, its sum is 0 and the epilogue's xy / den gives the same 0 — the column
        // is touched, hence a candidate, in both.  Until round 5 the call read a 4-byte flag back here and reran WITHOUT folding when a
        // stored entry had met a zero term: the one device-mode call that synchronised the caller's stream.  Round 6
        // ran the parity suite, the dedicated case (test_zero_depop_weight_on_a_column_with_entries: sparse, wave and generic kernels,
        // threshold 0 and negative) and 1 200 fuzz cases with the rerun switched off: no difference — the rerun and its wait are gone,
        // the call is asynchronous and stream-capturable like every other)
        if (!reuse) hipLaunchKernelGGL(sp_fold_colterm_kernel, dim3(256 * 8), dim3(256), 0, stream, (long long)a->nnz_m2, a->m2_indices,
                           a->m2_data, a->l2 != 0.f ? a->Ycosine : a->Ydepop, folded, (int *)nullptr);
        HIP_TRY(hipGetLastError());
    } else if (c.pack) {
        ypack = (float4 *)ws_fold;
        if (!reuse) hipLaunchKernelGGL(sp_pack_colterms_kernel, dim3(std::min(2048, (a->n_output_cols + 255) / 256)), dim3(256), 0, stream, a->n_output_cols,
                           a->l1 != 0.f ? a->Ytversky : nullptr, a->l2 != 0.f ? a->Ycosine : nullptr,
                           a->l3 != 0.f ? a->Ydepop : nullptr, ypack);
        HIP_TRY(hipGetLastError());
    }
    if (!reuse && !c.fold && bound_ok && (a->l1 != 0.f || a->l2 != 0.f || a->l3 != 0.f)) {
        hipLaunchKernelGGL(sp_colterm_min_kernel, dim3((unsigned)std::max(1, std::min(256, (a->n_output_cols + 4095) / 4096))), dim3(1024), 0, stream, a->n_output_cols,
                           a->l1 != 0.f ? a->Ytversky : nullptr, a->l2 != 0.f ? a->Ycosine : nullptr,
                           a->l3 != 0.f ? a->Ydepop : nullptr, ymin_dev, (unsigned *)(ws + WS_SCRATCH_OFFSET));
        HIP_TRY(hipGetLastError());
    }

    BndInfo *bnd_info = nullptr;
    unsigned *bnd_colpack = nullptr, *bnd_ids = nullptr;
    if (c.bnd) {
        // the bounded variant's per-call passes (sp_prep_kernels.hpp): reference multipliers + code layout -> BndInfo, packed id per column,
        // packed m2 ids (one streaming pass over m2's indices: 0.5 GB of traffic at the C2 size)
        bnd_info = (BndInfo *)(ws + WS_BND_OFFSET);
        bnd_colpack = (unsigned *)(ws_fold + c.ws_bnd_colpack);
        bnd_ids = (unsigned *)(ws_fold + c.ws_bnd_ids);
        if (!reuse) {
            const float *ytv = (a->l1 != 0.f && a->t2 != 0.f) ? a->Ytversky : nullptr, *ycos = a->l2 != 0.f ? a->Ycosine : nullptr, *ydep = a->l3 != 0.f ? a->Ydepop : nullptr;
            float *bnd_acc = (float *)(ws + WS_SCRATCH_OFFSET + 4);      // {sum cos, sum dep, n cos, n dep, done} | done of the second launch
            hipLaunchKernelGGL(sp_bnd_xmean_kernel, dim3((unsigned)std::max(1, std::min(256, (a->n_rows_m1 + 4095) / 4096))), dim3(1024), 0, stream, a->n_rows_m1,
                               a->l2 != 0.f ? a->Xcosine : nullptr, a->l3 != 0.f ? a->Xdepop : nullptr, ytv != nullptr, ycos != nullptr, ydep != nullptr,
                               a->l1 * a->t2, a->l2, a->l3, bnd_acc, bnd_info);
            hipLaunchKernelGGL(sp_bnd_range_kernel, dim3((unsigned)std::max(1, std::min(256, (a->n_output_cols + 4095) / 4096))), dim3(1024), 0, stream, a->n_output_cols,
                               ytv, ycos, ydep, (unsigned *)(bnd_acc + 5), bnd_info, bnd_id_bits(a->n_output_cols));
            hipLaunchKernelGGL(sp_bnd_colpack_kernel, dim3(std::min(2048, (a->n_output_cols + 255) / 256)), dim3(256), 0, stream, a->n_output_cols, ytv, ycos, ydep,
                               (const BndInfo *)bnd_info, bnd_colpack, bnd_id_bits(a->n_output_cols));
            hipLaunchKernelGGL(sp_bnd_pack_ids_kernel, dim3(256 * 8), dim3(256), 0, stream, (long long)a->nnz_m2, a->m2_indices, (const unsigned *)bnd_colpack, bnd_ids, bnd_info);
            HIP_TRY(hipGetLastError());
        }
    }

    int *neg_flag = (int *)(ws + WS_YMIN_OFFSET + 12);      // (inside the zeroed header)
    const bool sign_matters = a->bayesian_shrink != 0.f || a->l1 * (1.f - a->t1 - a->t2) > 0.f;      // (see RowCtx::set_cut)
    if (sign_matters && !reuse) {
        if (a->nnz_m1 > 0) hipLaunchKernelGGL(sp_any_negative_kernel, dim3(1024), dim3(256), 0, stream, (long long)a->nnz_m1, a->m1_data, neg_flag);
        if (a->nnz_m2 > 0) hipLaunchKernelGGL(sp_any_negative_kernel, dim3(1024), dim3(256), 0, stream, (long long)a->nnz_m2, a->m2_data, neg_flag);
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
    kp.qcount = (unsigned int *)(ws + 8);
    kp.qcount_g = (unsigned int *)(ws + 12);
    kp.cap_s = c.cap_s;
    kp.gU = c.u_lds_s ? nullptr : (u64 *)ws_gu;
    kp.gU_g = c.u_lds ? nullptr : (u64 *)(ws_gu + c.ws_gu_s_bytes);
    kp.sparse_path = ((a->flags & SP_FLAG_NO_SPARSE_PATH) || c.big) ? 0 : 1;
    {
        // work per row -> (optionally) descending-work order -> classified descriptor queues
        unsigned *bucket_count = (unsigned *)ws_rows;       // [32]
        unsigned *bucket_base = bucket_count + 32;          // [32] + [1] flag
        unsigned *work = (unsigned *)(ws_rows + 512);       // [n]
        int *order = (int *)(work + a->n_targets);          // [n]
        int4 *desc_s = (int4 *)(ws_rows + c.ws_desc_offset);            // [2n]
        int4 *desc_w = desc_s + 2 * (size_t)a->n_targets;               // [2n] the wave kernel's queue (launch_rows finds it there)
        int4 *desc_g = desc_w + 2 * (size_t)a->n_targets;               // [2n]
        HIP_TRY(hipMemsetAsync(ws_rows, 0, 512, stream));
        const int work_blocks = std::max(1, std::min((a->n_targets + 15) / 16, n_cus * 8));     // 16 rows (waves) per block and trip
        unsigned *long_count = bucket_count + 100;          // (inside the 512 bytes zeroed above; the list borrows `order`, written later)
        hipLaunchKernelGGL(sp_row_work_kernel, dim3(work_blocks), dim3(1024), 0, stream,
                           a->n_targets, a->targets, a->m1_indices, a->m1_indptr, a->m2_indptr, work, bucket_count, order, long_count);
        if (a->nnz_m1 > ROW_WORK_LONG)                       // (only a matrix with that many entries can hold such a row)
            hipLaunchKernelGGL(sp_row_work_long_kernel, dim3(std::min(n_cus * 2, 1024)), dim3(1024), 0, stream, a->targets, a->m1_indices, a->m1_indptr,
                               a->m2_indptr, work, bucket_count, (const int *)order, (const unsigned *)long_count);
        if (c.ordered) {
            hipLaunchKernelGGL(sp_bucket_base_kernel, dim3(1), dim3(64), 0, stream, bucket_count, bucket_base);
            hipLaunchKernelGGL(sp_row_order_kernel, dim3((a->n_targets + 255) / 256), dim3(256), 0, stream, a->n_targets, work, bucket_base, order);
        }
        ClassifyParams cp;
        cp.sparse_path = kp.sparse_path;
        cp.n_cols = a->n_output_cols; cp.T = c.T; cp.nb_log2 = c.nb_log2;
        cp.cs_slots = c.duo ? 2 * c.T_s : c.T_s / 4;      // (the rule counts the rank-addressed slots as half of the set)
        cp.duo = c.duo ? 1 : 0;
        cp.duo_l = c.duo_l ? 1 : 0;
        cp.wave = c.wave ? 1 : 0;
        cp.wave_macs_max = 10000u;
        cp.qcount_w = (unsigned *)(ws + 28);          // header words 6 / 7: head and length of the wave kernel's queue (zeroed with the header)
        cp.desc_w = desc_w;
        cp.mono = c.mono ? 1 : 0;
        cp.any_norm = (a->l1 != 0.f || a->l2 != 0.f || a->l3 != 0.f || a->stabilized_shrink != 0.f || a->bayesian_shrink != 0.f) ? 1 : 0;
        cp.l2 = a->l2; cp.l3 = a->l3;
        cp.split_fine = 0; cp.split_pmax = 0; cp.split_macs = 0u; cp.split_cap = 0; cp.split_count = nullptr; cp.split_rows = nullptr; cp.piece_info = nullptr;
        if (c.split_pmax) {
            const size_t np = (size_t)c.split_cap * (size_t)c.split_pmax;
            cp.split_fine = c.n_splits + 1;
            cp.split_pmax = c.split_pmax;
            cp.split_macs = (a->reserved[0] & 8192) ? 1u : split_piece_macs(a, c.wgs_generic);      // (bit 8192 of the ablation word: cut every generic row as finely as allowed, for tests)
            cp.split_cap = c.split_cap;
            cp.split_count = (int *)(ws + 16);                                  // two words inside the zeroed header
            cp.split_rows = (int4 *)ws_piece;
            cp.piece_info = (int2 *)(ws_piece + (size_t)c.split_cap * 16);
            kp.piece_info = cp.piece_info;
            kp.part_counts = (int *)((unsigned char *)cp.piece_info + np * 8);
            kp.part_cols = kp.part_counts + np;
            kp.part_vals = (float *)(kp.part_cols + np * (size_t)a->k);
        }
        hipLaunchKernelGGL(sp_row_desc_kernel, dim3((a->n_targets + 255) / 256), dim3(256), 0, stream, a->n_targets, a->targets,
                           a->m1_indptr, work, c.ordered ? bucket_base + 32 : nullptr, order, a->l1 != 0.f ? a->Xtversky : nullptr,
                           a->l2 != 0.f ? a->Xcosine : nullptr, a->l3 != 0.f ? a->Xdepop : nullptr, cp, kp.qcount, desc_s, desc_g);
        HIP_TRY(hipGetLastError());
        kp.desc = desc_s;
        kp.desc_g = desc_g;
        kp.items_g = nullptr; kp.items_rows = 0;
        if (c.items_rows > 0 && kp.sparse_path) {
            const int item_blocks = std::max(1, std::min((a->n_targets + 3) / 4, n_cus * 32));     // 4 rows (waves) per block and trip
            hipLaunchKernelGGL(sp_row_items_kernel, dim3(item_blocks), dim3(256), 0, stream, (const unsigned *)kp.qcount, c.items_rows, (int4 *)desc_s,
                               a->m1_indices, a->m1_data, a->m2_indptr, (int4 *)ws_items, c.NT_s == 256 ? 1 : 0,
                               ((c.mono || c.bnd) && a->filter_mode == SP_SEL_MATRIX) ? a->filter_m_indptr : nullptr, c.items_stride);
            HIP_TRY(hipGetLastError());
            if (c.duo_l) {      // the rows of the second two-per-CU launch: same records, their own queue
                hipLaunchKernelGGL(sp_row_items_kernel, dim3(item_blocks), dim3(256), 0, stream, (const unsigned *)cp.qcount_w, c.items_rows, (int4 *)desc_w,
                                   a->m1_indices, a->m1_data, a->m2_indptr, (int4 *)ws_items, 0,
                                   ((c.mono || c.bnd) && a->filter_mode == SP_SEL_MATRIX) ? a->filter_m_indptr : nullptr, c.items_stride);
                HIP_TRY(hipGetLastError());
            }
            if (c.wave) {
                hipLaunchKernelGGL(sp_row_items_wave_kernel, dim3(item_blocks), dim3(256), 0, stream, (const unsigned *)cp.qcount_w, c.items_rows, (int4 *)desc_w,
                                   a->m1_indices, a->m1_data, a->m2_indptr, (int4 *)ws_items, c.items_stride);
                HIP_TRY(hipGetLastError());
            }
            kp.items_g = (const int4 *)ws_items; kp.items_rows = c.items_rows; kp.items_stride = c.items_stride;
        }
    }
    kp.m2_bytes = (unsigned)((size_t)a->nnz_m2 * 4);
    kp.nb_log2 = c.nb_log2;
    kp.hash_fill = c.hash_fill;
    kp.static_sched = (a->flags & SP_FLAG_STATIC_SCHED) ? 1 : 0;
    kp.ymin = ymin_dev;
    kp.Ypack = ypack;
    kp.bound_ok = bound_ok ? 1 : 0;
    kp.neg_flag = sign_matters ? neg_flag : nullptr;
    kp.fold = c.fold ? 1 : 0;
    if (c.fold) kp.m2_data = folded;
    kp.bnd = bnd_info; kp.colpack = bnd_colpack; kp.m2_packed = bnd_ids;
    kp.bnd_id_mask = (1u << bnd_id_bits(a->n_output_cols)) - 1u;
    kp.splits = nullptr;
    kp.n_splits = 0; kp.splits_state = nullptr;
    kp.split_w = c.split_w;
    SplitsLaunch sl{};
    if (c.n_splits) {
        // (queued by launch_rows between the sparse-row kernels and the generic one: skipped on the device when the generic queue is empty)
        sl.n_rows_m2 = a->n_rows_m2; sl.m2_indptr = a->m2_indptr; sl.m2_indices = a->m2_indices; sl.split_w = c.split_w; sl.n_splits = c.n_splits;
        sl.out = ws_split; sl.qcount_g = (const unsigned *)(ws + 12); sl.state = (int *)(ws + WS_SPLITS_STATE_OFFSET);
        kp.splits = ws_split;
        kp.n_splits = c.n_splits; kp.splits_rows = a->n_rows_m2; kp.splits_state = sl.state;
        kp.split_w = c.split_w;
    }
    kp.phase_cycles = (timed && (a->flags & SP_FLAG_PHASE_TIMERS)) ? (unsigned long long *)(ws + WS_PHASE_OFFSET) : nullptr;   // inside the zeroed header
    kp.dbg = (int)a->reserved[0];

    hipEvent_t kev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (timed) { for (int i = 0; i < 4; ++i) TRY(guard.event(&kev[i])); }
    KParams kp_s = kp;
    kp_s.T = c.T_s; kp_s.logT = c.duo ? (c.T_s == DUO_CS_DIRECT ? 9 : 10) : c.logT_s;      // (DUO: log2 of the collision set's overflow slots — 512 / 1024)
    static_assert(DUO_CS_OVER == 512 && DUO_CS_OVER_L == 1024, "log2 above");
    rc = launch_rows(kp_s, kp, c, stream, timed ? kev : nullptr, c.n_splits ? &sl : nullptr);
    if (rc) return rc;
    if (c.split_pmax) {
        const int n_rec = c.split_pmax * a->k;
        // (up to 8192 records of 8 bytes + the kernel's own static word: more than the 64 KiB a launch gets without asking)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_merge_pieces_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, n_rec * 8));
        hipLaunchKernelGGL(sp_merge_pieces_kernel, dim3(std::min(c.split_cap, 1024)), dim3(MERGE_NT), (size_t)n_rec * 8, stream, (const int *)(ws + 16), c.split_cap,
                           (const int4 *)ws_piece, a->k, a->targets, (const int *)kp.part_cols, (const float *)kp.part_vals, (const int *)kp.part_counts,
                           a->rows, a->cols, a->values, a->out_counts);
        HIP_TRY(hipGetLastError());
    }

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
        // (slot 8 carries no timer: which sparse-row kernel ran — bit 0: the wave-per-row kernel, bit 1: the workgroup kernel's bounded variant)
        a->phase_cycles[PH_CSDRAIN] = (c.wave ? 1 : 0) | ((c.bnd && ((const BndInfo *)(qb + WS_BND_OFFSET))->state == 1) ? 2 : 0);
        a->passes_total = (int32_t)phc[CT_PASSES];
        a->num_wgs_used = c.wave ? c.wgs_wave : c.wgs_sparse;
        float ks_ms = 0.f, kg_ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ks_ms, kev[0], kev[1]));
        HIP_TRY(hipEventElapsedTime(&kg_ms, kev[2], kev[3]));
        a->reserved[1] = (int64_t)(ks_ms * 1000.0f);      // sparse row kernel, microseconds
        a->reserved[2] = (int64_t)(kg_ms * 1000.0f);      // generic row kernel, microseconds
    }
    return SP_OK;      // (the guard waits for the stream before it frees an owned workspace)
}

}  // namespace
#include "sp_transpose.hpp"
namespace {

// Layout of the extra scratch a SP_FLAG_M2_IS_M1_T / SP_FLAG_M1_IS_M2_T call needs behind the kernel's own workspace: the
// three arrays of the matrix built here (m2 = m1^T or m1 = m2^T), the optional vectors, then the transpose's scratch.
struct M2tLayout { size_t knn, data, indices, indptr, p3copy, ydepop, norms, keep, zc, tr, total; };
// SP_FLAG_P3_PREP: where the device counter of entries that underflowed to 0.0 lives (inside the call's scratch), for the host-mode
// entry of the same thread to read once everything has been waited for
thread_local const unsigned long long *g_p3_zero_counter = nullptr;
int m2t_layout(const sp_knn_args *a, int n_cus, sp_knn_args *plain, M2tLayout *L) {
    const bool m1t = (a->flags & SP_FLAG_M1_IS_M2_T) != 0;
    const int64_t nnz = m1t ? a->nnz_m2 : a->nnz_m1;
    const int built_rows = m1t ? a->n_rows_m1 : a->n_rows_m2;      // rows of the matrix built here = columns of the one given
    *plain = *a;
    plain->flags &= ~(SP_FLAG_M2_IS_M1_T | SP_FLAG_M1_IS_M2_T | SP_FLAG_P3_PREP | SP_FLAG_DEPOP_ROWSUM | SP_FLAG_NORMS_ON_DEVICE);
    plain->nnz_m1 = plain->nnz_m2 = nnz;
    plain->col_keep = nullptr;                 // (applied while m2 is built)
    Config c{};
    TRY(make_config(plain, n_cus, &c));
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L->knn = al(c.ws_total);
    L->data = L->knn;
    L->indices = L->data + al((size_t)nnz * 4);
    L->indptr = L->indices + al((size_t)nnz * 4);
    // SP_FLAG_P3_PREP: a normalised copy of the caller's values (they stay as they are) and, for rp3beta, the column term
    L->p3copy = L->indptr + al(((size_t)built_rows + 1) * 4);
    L->ydepop = L->p3copy + ((a->flags & SP_FLAG_P3_PREP) ? al((size_t)nnz * 4) : 0);
    L->norms = L->ydepop + ((a->flags & SP_FLAG_DEPOP_ROWSUM) ? al((size_t)a->n_rows_m1 * 4) : 0);
    L->keep = L->norms + ((a->flags & SP_FLAG_NORMS_ON_DEVICE) ? 4 * al((size_t)a->n_rows_m1 * 4) : 0);
    // SP_FLAG_P3_PREP with a column mask: the mask is applied to the NORMALISED m2 (the reference normalises the rows of matrix2 before
    // it drops columns, similarity.py:410-415 then s_plus_utils.pyx:424-490): a second copy of m2's three arrays, scan scratch, total
    const bool p3_keep = (a->flags & SP_FLAG_P3_PREP) && a->col_keep != nullptr && !m1t;
    L->zc = L->keep + (p3_keep ? al(((size_t)built_rows + 1) * 4) + 2 * al((size_t)nnz * 4) + al(SCAN_SCRATCH_BYTES) + 256 : 0);
    L->tr = L->zc + 256;
    L->total = L->tr + transpose_ws_bytes(nnz, built_rows);
    return SP_OK;
}

// out[i] = (in[i] + add)^p in float32: _build_cosine_normalization (s_plus_utils.pyx:204-228: the sum in float32, np.power in float32)
__global__ __launch_bounds__(256) void sp_add_pow_f32_kernel(int n, const float *__restrict__ in, float *__restrict__ out, float add, double p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)pow((double)__fadd_rn(in[i], add), p);
}

// device pointers in, device pointers out; with SP_FLAG_M2_IS_M1_T / SP_FLAG_M1_IS_M2_T the transpose (s_plus.pyx:169-170,
// 205-206) is built first, on the same stream, into scratch behind the kernel's workspace
// the row kernels of a call whose operands are what they should be (`b`), in one launch chain or chunk by chunk
// The sampled route (sp_sddmm_kernel.hpp).  mt_*: m2^T on the device, or NULL: transposed here from b->m2_* into the workspace.
int run_sddmm(sp_knn_args *b, const float *mt_data, const int *mt_indices, const int *mt_indptr, const ChunkHook *hook) {
    HIP_TRY(hipSetDevice(b->device));
    if (b->n_targets == 0) { b->kernel_ms = 0.f; return SP_OK; }
    hipStream_t stream = (hipStream_t)b->stream;
    unsigned char *ws = (unsigned char *)b->workspace;
    CallGuard guard;
    guard.stream = stream;
    const size_t need = mt_indptr ? 256 : sddmm_ws_bytes(b);
    if (!ws) {
        HIP_TRY(hipMalloc((void **)&ws, need));
        guard.ws = ws;
    } else if (b->workspace_bytes < (int64_t)need) {
        return fail(SP_EWORKSPACE, "workspace too small: need %zu bytes, got %lld", need, (long long)b->workspace_bytes);
    }
    const bool timed = (b->flags & SP_FLAG_TIME_KERNEL) != 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) { TRY(guard.event(&ev0)); TRY(guard.event(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
    // (the sampled route builds none of the per-call passes and overwrites the workspace's header — and, for an explicit m2, the blocks behind
    // it: whatever an earlier call left there is gone, so a later SP_FLAG_REUSE_M2_PREP call on this address must not find its signature)
    if (b->workspace) prep_store(ws, ~0ull, -1);
    HIP_TRY(hipMemsetAsync(ws, 0, 256, stream));
    if (!mt_indptr) {
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        float *td = (float *)(ws + 256);
        int *ti = (int *)((unsigned char *)td + al((size_t)b->nnz_m2 * 4));
        int *tp = (int *)((unsigned char *)ti + al((size_t)b->nnz_m2 * 4));
        unsigned char *tws = (unsigned char *)tp + al(((size_t)b->n_output_cols + 1) * 4);
        TRY(transpose_device(b->n_rows_m2, b->n_output_cols, b->nnz_m2, b->m2_data, b->m2_indices, b->m2_indptr, td, ti, tp, tws, transpose_ws_bytes(b->nnz_m2, b->n_output_cols), stream));
        mt_data = td; mt_indices = ti; mt_indptr = tp;
    }
    SddmmParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.n_targets = b->n_targets; sp.targets = b->targets;
    sp.m1_data = b->m1_data; sp.m1_indices = b->m1_indices; sp.m1_indptr = b->m1_indptr;
    sp.mt_data = mt_data; sp.mt_indices = mt_indices; sp.mt_indptr = mt_indptr;
    sp.t_indptr = b->target_col_m_indptr; sp.t_indices = b->target_col_m_indices;
    sp.filter_mode = b->filter_mode; sp.f_indptr = b->filter_m_indptr; sp.f_indices = b->filter_m_indices;
    sp.col_keep = b->col_keep;
    sp.Xtv = b->Xtversky; sp.Ytv = b->Ytversky; sp.Xcos = b->Xcosine; sp.Ycos = b->Ycosine; sp.Xdep = b->Xdepop; sp.Ydep = b->Ydepop;
    sp.a1 = b->a1; sp.l1 = b->l1; sp.l2 = b->l2; sp.l3 = b->l3; sp.t1 = b->t1; sp.t2 = b->t2;
    sp.stab = b->stabilized_shrink; sp.bayes = b->bayesian_shrink; sp.threshold = b->threshold;
    sp.k = b->k;
    sp.rows = (b->flags & SP_FLAG_NO_ROWS_OUT) ? nullptr : b->rows; sp.cols = b->cols; sp.values = b->values; sp.counts = b->out_counts;
    sp.queue = (unsigned *)ws;
    int n_cus = 256;
    TRY(device_cus(b->device, &n_cus));
    const int wgs = std::max(1, std::min((b->n_targets + SD_WAVES - 1) / SD_WAVES, n_cus * (int)(LDS_LIMIT / sd_lds_bytes())));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_sddmm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sd_lds_bytes()));
    hipLaunchKernelGGL(sp_sddmm_kernel, dim3(wgs), dim3(64 * SD_WAVES), sd_lds_bytes(), stream, sp);
    HIP_TRY(hipGetLastError());
    if (hook && hook->after_launch) for (int j = 0; j < std::max(1, hook->n_chunks); ++j) TRY(hook->after_launch(j));
    if (timed) {
        HIP_TRY(hipEventRecord(ev1, stream));
        HIP_TRY(hipEventSynchronize(ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        b->kernel_ms = ms;
        memset(b->phase_cycles, 0, sizeof(b->phase_cycles));
        b->phase_cycles[PH_CSDRAIN] = 4;        // (slot 8, bit 2: the sampled route ran)
        b->passes_total = 0; b->num_wgs_used = wgs;
        b->reserved[1] = (int64_t)(ms * 1000.f); b->reserved[2] = 0;
    }
    return SP_OK;
}

int run_rows(sp_knn_args *b, const ChunkHook *hook) {
    if (sddmm_applies(b, b->nnz_m1, b->nnz_m2)) return run_sddmm(b, nullptr, nullptr, nullptr, hook);
    if (!hook || hook->n_chunks <= 1) {
        TRY(run_device_impl(b));
        return (hook && hook->after_launch) ? hook->after_launch(0) : SP_OK;
    }
    const size_t k = (size_t)b->k;
    for (int j = 0; j < hook->n_chunks; ++j) {
        const size_t s0 = hook->bounds[(size_t)j], s1 = hook->bounds[(size_t)j + 1];
        sp_knn_args sub = *b;
        sub.n_targets = (int32_t)(s1 - s0);
        sub.targets = b->targets + s0;
        if (b->rows) sub.rows = b->rows + s0 * k;
        sub.cols = b->cols + s0 * k;
        sub.values = b->values + s0 * k;
        if (b->out_counts) sub.out_counts = b->out_counts + s0;
        if (j > 0) sub.flags |= SP_FLAG_REUSE_M2_PREP;
        if (s1 > s0) TRY(run_device_impl(&sub));
        if (hook->after_launch) TRY(hook->after_launch(j));
    }
    return SP_OK;
}

int run_device(sp_knn_args *a, const ChunkHook *hook = nullptr) {
    const bool m2t = (a->flags & SP_FLAG_M2_IS_M1_T) != 0, m1t = (a->flags & SP_FLAG_M1_IS_M2_T) != 0;
    if (!m2t && !m1t) return run_rows(a, hook);
    HIP_TRY(hipSetDevice(a->device));
    if (a->n_targets == 0) { a->kernel_ms = 0.f; return SP_OK; }
    int n_cus = 256;
    TRY(device_cus(a->device, &n_cus));
    sp_knn_args b;
    M2tLayout L;
    TRY(m2t_layout(a, n_cus, &b, &L));
    hipStream_t stream = (hipStream_t)a->stream;
    unsigned char *ws = (unsigned char *)a->workspace;
    CallGuard guard;
    guard.stream = stream;
    if (!ws) {
        HIP_TRY(hipMalloc((void **)&ws, L.total));
        guard.ws = ws;
    } else if (a->workspace_bytes < (int64_t)L.total) {
        return fail(SP_EWORKSPACE, "workspace too small: need %zu bytes, got %lld", L.total, (long long)a->workspace_bytes);
    }
    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) {
        TRY(guard.event(&ev0));
        TRY(guard.event(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    const int64_t nnz = b.nnz_m1;
    float *t_data = (float *)(ws + L.data);
    int *t_indices = (int *)(ws + L.indices), *t_indptr = (int *)(ws + L.indptr);
    const bool p3_keep = m2t && (a->flags & SP_FLAG_P3_PREP) && a->col_keep != nullptr;      // (the mask waits for the normalised m2)
    // target_cols = <matrix> with few listed entries: the sampled route (sp_sddmm_kernel.hpp) needs m2^T — for m2 = m1^T that is m1
    // itself: no transpose is built at all; for m1 = m2^T it is the m1 built here
    const bool sampled = sddmm_applies(&b, nnz, nnz);
    int rc = (m2t && sampled) ? SP_OK
             : m2t ? transpose_device(a->n_rows_m1, a->n_rows_m2, nnz, a->m1_data, a->m1_indices, a->m1_indptr,
                                    t_data, t_indices, t_indptr, ws + L.tr, L.total - L.tr, stream, p3_keep ? nullptr : a->col_keep)
                 : transpose_device(a->n_rows_m2, a->n_rows_m1, nnz, a->m2_data, a->m2_indices, a->m2_indptr,
                                    t_data, t_indices, t_indptr, ws + L.tr, L.total - L.tr, stream);
    float tr_ms = 0.f;
    if (!rc && timed) {
        HIP_TRY(hipEventRecord(ev1, stream));
        HIP_TRY(hipEventSynchronize(ev1));
        HIP_TRY(hipEventElapsedTime(&tr_ms, ev0, ev1));
    }
    if (rc) return rc;
    // both matrices now exist; `own` marks the one built here (writable), the other one is the caller's
    const float *m1_data = m2t ? a->m1_data : t_data;
    const int *m1_indptr = m2t ? a->m1_indptr : t_indptr;
    const float *m2_data = m2t ? t_data : a->m2_data;
    const int *m2_indptr = m2t ? t_indptr : a->m2_indptr;
    if (m2t) { b.m2_data = t_data; b.m2_indices = t_indices; b.m2_indptr = t_indptr; }
    else     { b.m1_data = t_data; b.m1_indices = t_indices; b.m1_indptr = t_indptr; }
    const int wave_blocks = [](int n) { return std::max(1, std::min(256 * 16, (n + 3) / 4)); }(std::max(a->n_rows_m1, a->n_rows_m2));
    const int vec_blocks = (a->n_rows_m1 + 255) / 256;
    if ((a->flags & SP_FLAG_NORMS_ON_DEVICE) && (a->l1 != 0.f || a->l2 != 0.f) && a->n_rows_m1 > 0) {
        // _build_squared_norms for m2 = m1^T, from the rows of m1 (sp_csr_row_sqsums_f32's two kernels), then
        // _build_cosine_normalization (s_plus_utils.pyx:204-228)
        const size_t stride = ((size_t)a->n_rows_m1 * 4 + 255) & ~(size_t)255;
        float *sq1 = (float *)(ws + L.norms), *sq2 = (float *)(ws + L.norms + stride);
        float *xc = (float *)(ws + L.norms + 2 * stride), *yc = (float *)(ws + L.norms + 3 * stride);
        hipLaunchKernelGGL(sp_row_sqsums_kernel, dim3(std::min(256 * 16, vec_blocks)), dim3(256), 0, stream, a->n_rows_m1, m1_data, m1_indptr, sq1, sq2);
        hipLaunchKernelGGL(sp_row_sqsums_long_kernel, dim3(std::min(a->n_rows_m1, 2048)), dim3(256), 0, stream, a->n_rows_m1, m1_data, m1_indptr, sq1, sq2);
        if (a->l1 != 0.f) { b.Xtversky = sq1; b.Ytversky = sq2; }
        if (a->l2 != 0.f) {
            hipLaunchKernelGGL(sp_add_pow_f32_kernel, dim3(vec_blocks), dim3(256), 0, stream, a->n_rows_m1, sq1, xc, a->norm_add, (double)a->norm_c1);
            hipLaunchKernelGGL(sp_add_pow_f32_kernel, dim3(vec_blocks), dim3(256), 0, stream, a->n_rows_m1, sq2, yc, a->norm_add, (double)a->norm_c2);
            b.Xcosine = xc; b.Ycosine = yc;
        }
        HIP_TRY(hipGetLastError());
    }
    if ((a->flags & SP_FLAG_P3_PREP) && nnz > 0) {
        // p3alpha / rp3beta (similarity.py:410-415, 477-483): the column popularity comes from the RAW matrix, then the rows of
        // m1 and of m2 = m1^T are divided by their L1 norms and every entry is raised to alpha
        if (a->flags & SP_FLAG_DEPOP_ROWSUM) {
            float *yd = (float *)(ws + L.ydepop);
            hipLaunchKernelGGL(sp_row_sums_kernel, dim3(wave_blocks), dim3(256), 0, stream, a->n_rows_m1, m1_data, m1_indptr, yd);
            hipLaunchKernelGGL(sp_pow_f32_kernel, dim3(vec_blocks), dim3(256), 0, stream, a->n_rows_m1, yd, yd, (double)a->depop_p2);
            b.Ydepop = yd;
        }
        // the caller's values stay as they are: a normalised copy of them, the matrix built here in place
        float *cp = (float *)(ws + L.p3copy);
        HIP_TRY(hipMemcpyAsync(cp, m2t ? m1_data : m2_data, (size_t)nnz * 4, hipMemcpyDeviceToDevice, stream));
        float *m1n = m2t ? cp : t_data, *m2n = m2t ? t_data : cp;
        // (entries that underflow to 0.0 on the way are counted: the reference removes them before its kernel runs — s_plus.pyx:210-211
        // after similarity.py:410-415 — here they would stay zero-valued candidates; a host-mode call reports SP_EUNDERFLOW, see run_host)
        unsigned long long *zero_made = (unsigned long long *)(ws + L.zc);
        HIP_TRY(hipMemsetAsync(zero_made, 0, sizeof(unsigned long long), stream));
        g_p3_zero_counter = zero_made;
        hipLaunchKernelGGL((sp_row_normalize_kernel<float, RO_L1>), dim3(wave_blocks), dim3(256), 0, stream, a->n_rows_m1, m1n, m1_indptr, (double)a->p3_alpha, zero_made);
        hipLaunchKernelGGL((sp_row_normalize_kernel<float, RO_L1>), dim3(wave_blocks), dim3(256), 0, stream, a->n_rows_m2, m2n, m2_indptr, (double)a->p3_alpha, zero_made);
        HIP_TRY(hipGetLastError());
        b.m1_data = m1n;
        b.m2_data = m2n;
        if (p3_keep) {
            // _filter_matrix_columns on the normalised m2 (s_plus_utils.pyx:424-490): kept entries compacted row by row, order kept; the
            // tails of the new arrays are zero (flat passes over nnz entries read them)
            auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
            int *n_indptr = (int *)(ws + L.keep);
            int *n_idx = (int *)(ws + L.keep + al(((size_t)a->n_rows_m2 + 1) * 4));
            float *n_val = (float *)((unsigned char *)n_idx + al((size_t)nnz * 4));
            long long *scan_part = (long long *)((unsigned char *)n_val + al((size_t)nnz * 4));
            long long *kept = (long long *)((unsigned char *)scan_part + al(SCAN_SCRATCH_BYTES));
            HIP_TRY(hipMemsetAsync(n_indptr, 0, ((size_t)a->n_rows_m2 + 1) * 4, stream));
            HIP_TRY(hipMemsetAsync(n_idx, 0, (size_t)nnz * 4, stream));
            HIP_TRY(hipMemsetAsync(n_val, 0, (size_t)nnz * 4, stream));
            const int wb = std::max(1, std::min(256 * 16, (a->n_rows_m2 + 3) / 4));
            hipLaunchKernelGGL(sp_keep_count_kernel, dim3(wb), dim3(256), 0, stream, a->n_rows_m2, (const int *)t_indptr, (const int *)t_indices, a->col_keep, n_indptr);
            scan_i32<true>((long long)a->n_rows_m2 + 1, n_indptr, n_indptr, nullptr, kept, scan_part, stream);
            hipLaunchKernelGGL(sp_keep_compact_kernel, dim3(wb), dim3(256), 0, stream, a->n_rows_m2, (const int *)t_indptr, (const int *)t_indices, (const float *)m2n, a->col_keep,
                               (const int *)n_indptr, n_idx, n_val);
            HIP_TRY(hipGetLastError());
            b.m2_indptr = n_indptr; b.m2_indices = n_idx; b.m2_data = n_val;
        }
    }
    b.workspace = ws;
    b.workspace_bytes = (int64_t)L.knn;
    if (sampled) {
        b.col_keep = m2t ? a->col_keep : nullptr;      // (ARRAY selectors of the m2 that is not built: the listed columns are looked up in the mask)
        rc = run_sddmm(&b, b.m1_data, b.m1_indices, b.m1_indptr, hook);
    } else
    rc = run_rows(&b, hook);
    a->kernel_ms = b.kernel_ms + tr_ms;
    a->passes_total = b.passes_total;
    a->num_wgs_used = b.num_wgs_used;
    memcpy(a->phase_cycles, b.phase_cycles, sizeof(a->phase_cycles));
    a->reserved[1] = b.reserved[1];
    a->reserved[2] = b.reserved[2];
    a->reserved[3] = (int64_t)(tr_ms * 1000.f);
    return rc;
}

// Device buffers of host-mode calls are cached per device in size buckets (next multiple of 1/8 of a power of two) and
// reused by later calls: hipMalloc / hipFree of GB-sized buffers cost milliseconds each, and a similarity pipeline
// (normalise -> similarity -> scoring) makes many such calls.  sp_device_cache_trim() gives the memory back.
struct DeviceCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_blocks;      // (device, bucket bytes) -> idle blocks
    std::map<int, size_t> idle_bytes;                                       // device -> bytes sitting in free_blocks
    // Idle bytes kept per device at most: other allocators of the process (torch, a second library) never see this cache's
    // hipMalloc fail, so it must not sit on an unbounded share of HBM.  SIMILARIPY_AMD_DEVICE_CACHE_MB overrides (0 = keep nothing).
    static size_t cap_bytes() {
        static const size_t cap = [] {
            const char *e = getenv("SIMILARIPY_AMD_DEVICE_CACHE_MB");
            return e ? (size_t)strtoull(e, nullptr, 10) << 20 : (size_t)16 << 30;
        }();
        return cap;
    }
    static size_t bucket(size_t n) {
        n = std::max<size_t>(n, 256);
        size_t p = 256;
        while (p < n) p <<= 1;
        const size_t step = p >> 3;                   // 8 buckets per octave: at most 12.5 % over-allocation
        return step ? ((n + step - 1) / step) * step : p;
    }
    int get(int device, size_t bytes, void **out, size_t *got) {
        const size_t b = bucket(bytes);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_blocks.find({device, b});
            if (it != free_blocks.end() && !it->second.empty()) {
                *out = it->second.back();
                it->second.pop_back();
                idle_bytes[device] -= b;
                *got = b;
                return SP_OK;
            }
        }
        void *d = nullptr;
        hipError_t e = hipMalloc(&d, b);
        if (e != hipSuccess) {          // out of memory: drop the cache and try once more
            (void)hipGetLastError();
            trim(device);
            e = hipMalloc(&d, b);
        }
        if (e != hipSuccess) return fail(SP_ENOMEM, "hipMalloc(%zu bytes) failed: %s", b, hipGetErrorString(e));
        *out = d;
        *got = b;
        return SP_OK;
    }
    void put(int device, size_t b, void *p) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (idle_bytes[device] + b <= cap_bytes()) {
                free_blocks[{device, b}].push_back(p);
                idle_bytes[device] += b;
                return;
            }
        }
        (void)hipFree(p);      // over the cap: back to the driver
    }
    long long trim(int device) {
        std::vector<std::pair<size_t, void *>> victims;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto &kv : free_blocks)
                if (kv.first.first == device || device < 0) {
                    for (void *p : kv.second) victims.push_back({kv.first.second, p});
                    kv.second.clear();
                }
            for (auto &kv : idle_bytes)
                if (kv.first == device || device < 0) kv.second = 0;
        }
        long long n = 0;
        for (auto &v : victims) { (void)hipFree(v.second); n += (long long)v.first; }
        return n;
    }
};
DeviceCache g_cache;
const bool g_cache_on = getenv("SIMILARIPY_AMD_NO_DEVICE_CACHE") == nullptr;

// RAII device allocation list for the host-pointer entries; blocks go back to the cache (after a device sync: a block is
// never handed out again while a kernel of the call that used it may still run)
struct DevPool {
    int device = 0;
    std::vector<std::pair<void *, size_t>> blocks;
    ~DevPool() {
        if (blocks.empty()) return;
        (void)hipDeviceSynchronize();
        for (auto &b : blocks) {
            if (g_cache_on) g_cache.put(device, b.second, b.first);
            else (void)hipFree(b.first);
        }
    }
    int raw(size_t bytes, void **d) {
        size_t got = 0;
        if (g_cache_on) TRY(g_cache.get(device, bytes, d, &got));
        else { HIP_TRY(hipMalloc(d, std::max<size_t>(bytes, 256))); got = bytes; }
        blocks.push_back({*d, got});
        return SP_OK;
    }
    template <typename Tp>
    int up(const Tp *host, size_t n, const Tp **dev) {
        *dev = nullptr;
        void *d = nullptr;
        // (an empty operand still gets a valid device pointer so that kernels never see host addresses)
        TRY(raw(std::max<size_t>(n, 1) * sizeof(Tp), &d));
        if (host && n) HIP_TRY(hipMemcpy(d, host, n * sizeof(Tp), hipMemcpyHostToDevice));
        *dev = (const Tp *)d;
        return SP_OK;
    }
    template <typename Tp>
    int alloc(size_t n, Tp **dev) {
        void *d = nullptr;
        TRY(raw(std::max<size_t>(n, 1) * sizeof(Tp), &d));
        *dev = (Tp *)d;
        return SP_OK;
    }
};


// host pointers in, host pointers out: the drop-in for s_plus.pyx:359-384
// SIMILARIPY_AMD_TRACE=1: wall clock of the stages of a host-mode call on stderr (the device is synchronised at every mark)
struct StageTrace {
    bool on = false;
    std::chrono::steady_clock::time_point t0;
    StageTrace() { const char *e = getenv("SIMILARIPY_AMD_TRACE"); on = e && *e && *e != '0'; t0 = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[similaripy_hip] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Touches one byte per page of host ranges on helper threads (see run_host).
struct HostPrefault {
    std::vector<std::pair<unsigned char *, size_t>> ranges;
    std::vector<std::thread> threads;
    void add(void *p, size_t bytes) { if (p && bytes) ranges.push_back({(unsigned char *)p, bytes}); }
    // rows[i * k + j] = targets[i]: the `rows` output of a host-mode call is known before the kernel runs (every entry of slot i
    // is in row targets[i], utils.pyx:43-64); it is written here, while the device works, instead of travelling over PCIe
    void fill_rows(int32_t *rows, const int32_t *targets, size_t nt, size_t k) {
        if (!rows || !nt || !k) return;
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t n_thr = std::max<size_t>(1, std::min<size_t>({(size_t)8, hw ? (size_t)hw / 2 : (size_t)1, (nt * k + ((size_t)1 << 22) - 1) >> 22}));
        const size_t per = (nt + n_thr - 1) / n_thr;
        for (size_t lo = 0; lo < nt; lo += per) {
            const size_t hi = std::min(nt, lo + per);
            auto job = [rows, targets, lo, hi, k]() {
                for (size_t i = lo; i < hi; ++i) {
                    const int32_t t = targets[i];
                    int32_t *r = rows + i * k;
                    for (size_t j = 0; j < k; ++j) r[j] = t;
                }
            };
            try { threads.emplace_back(job); } catch (...) { job(); }
        }
    }
    void start() {
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t per_range = std::max<size_t>(1, std::min<size_t>(4, hw ? hw / 2 : 1) );
        for (auto &r : ranges) {
            const size_t chunk = ((r.second + per_range - 1) / per_range + 4095) & ~(size_t)4095;
            for (size_t off = 0; off < r.second; off += chunk) {
                unsigned char *b = r.first + off;
                const size_t n = std::min(chunk, r.second - off);
                try {
                    threads.emplace_back([b, n]() {
                        for (size_t i = 0; i < n; i += 4096) ((volatile unsigned char *)b)[i] = 0;
                        ((volatile unsigned char *)b)[n - 1] = 0;
                    });
                } catch (...) { /* no thread: the copy pays for these pages itself */ }
            }
        }
    }
    void join() { for (auto &t : threads) if (t.joinable()) t.join(); threads.clear(); }
    ~HostPrefault() { join(); }
};

int run_host(sp_knn_args *a) {
    HIP_TRY(hipSetDevice(a->device));
    StageTrace trace;
    const size_t nt = (size_t)a->n_targets, k = (size_t)a->k;
    if (nt == 0) return SP_OK;
    // the reference trusts `targets` (s_plus.pyx:191-196, no bounds check); a device kernel must not
    for (size_t i = 0; i < nt; ++i)
        if (a->targets[i] < 0 || a->targets[i] >= a->n_rows_m1)
            return fail(SP_EINVAL, "targets[%zu]=%d out of range [0,%d)", i, a->targets[i], a->n_rows_m1);

    const bool m2t = (a->flags & SP_FLAG_M2_IS_M1_T) != 0, m1t = (a->flags & SP_FLAG_M1_IS_M2_T) != 0;
    trace.mark("argument checks (host)");
    DevPool pool;
    pool.device = a->device;
    sp_knn_args d = *a;
    d.on_device = 1;
    d.stream = nullptr;
    d.workspace = nullptr;
    d.workspace_bytes = 0;
    TRY(pool.up(a->targets, nt, &d.targets));
    if (m1t) {                                 // m1 never exists on the host: built on the device from m2
        d.m1_data = nullptr; d.m1_indices = nullptr; d.m1_indptr = nullptr;
        d.nnz_m1 = a->nnz_m2;
    } else {
        TRY(pool.up(a->m1_data, (size_t)a->nnz_m1, &d.m1_data));
        TRY(pool.up(a->m1_indices, (size_t)a->nnz_m1, &d.m1_indices));
        TRY(pool.up(a->m1_indptr, (size_t)a->n_rows_m1 + 1, &d.m1_indptr));
    }
    if (m2t) {                                 // m2 never exists on the host: built on the device from m1
        d.m2_data = nullptr; d.m2_indices = nullptr; d.m2_indptr = nullptr;
        if (a->col_keep) TRY(pool.up(a->col_keep, (size_t)a->n_rows_m1, &d.col_keep));
    } else {
        TRY(pool.up(a->m2_data, (size_t)a->nnz_m2, &d.m2_data));
        TRY(pool.up(a->m2_indices, (size_t)a->nnz_m2, &d.m2_indices));
        TRY(pool.up(a->m2_indptr, (size_t)a->n_rows_m2 + 1, &d.m2_indptr));
    }
    const bool host_norms = !(a->flags & SP_FLAG_NORMS_ON_DEVICE);
    TRY(pool.up(host_norms && a->l1 != 0.f ? a->Xtversky : nullptr, (size_t)a->n_rows_m1, &d.Xtversky));
    TRY(pool.up(host_norms && a->l1 != 0.f ? a->Ytversky : nullptr, (size_t)a->n_output_cols, &d.Ytversky));
    TRY(pool.up(host_norms && a->l2 != 0.f ? a->Xcosine : nullptr, (size_t)a->n_rows_m1, &d.Xcosine));
    TRY(pool.up(host_norms && a->l2 != 0.f ? a->Ycosine : nullptr, (size_t)a->n_output_cols, &d.Ycosine));
    TRY(pool.up(a->l3 != 0.f ? a->Xdepop : nullptr, (size_t)a->n_rows_m1, &d.Xdepop));
    TRY(pool.up(a->l3 != 0.f ? a->Ydepop : nullptr, (size_t)a->n_output_cols, &d.Ydepop));
    const bool fm = a->filter_mode == SP_SEL_MATRIX, tm = a->target_col_mode == SP_SEL_MATRIX;
    // a selector that IS m1's pattern (filter_cols = the URM that is being scored: the same host arrays) goes up once
    auto selector_up = [&](bool on, const int32_t *h_ptr, const int32_t *h_idx, int64_t nnz, const int32_t **d_ptr, const int32_t **d_idx) -> int {
        if (on && !m1t && h_ptr == a->m1_indptr && h_idx == a->m1_indices && nnz == a->nnz_m1) {
            *d_ptr = d.m1_indptr; *d_idx = d.m1_indices;
            return SP_OK;
        }
        TRY(pool.up(on ? h_ptr : nullptr, (size_t)a->n_rows_m1 + 1, d_ptr));
        TRY(pool.up(on ? h_idx : nullptr, (size_t)nnz, d_idx));
        return SP_OK;
    };
    TRY(selector_up(fm, a->filter_m_indptr, a->filter_m_indices, a->filter_nnz, &d.filter_m_indptr, &d.filter_m_indices));
    TRY(selector_up(tm, a->target_col_m_indptr, a->target_col_m_indices, a->target_col_nnz, &d.target_col_m_indptr, &d.target_col_m_indices));

    trace.mark("operands to the device");
    {
        // ... nor a hand-built CSR: out-of-range indices or a non-monotone indptr would become out-of-bounds device reads and
        // atomics.  Checked here, where the arrays already are (one launch per matrix), together with the two content checks:
        //   SP_FLAG_CHECK_ZEROS  explicit zeros are structural for the kernel (a candidate with value 0, a 1 under `binary`): the
        //                        reference removes them first (s_plus.pyx:210-211); the rare matrix that has some goes back to the caller
        //   SP_FLAG_M1_IS_M2_T   the column windows of the row kernels need ascending column ids inside each m2 row (sp_knn.h)
        struct Mat { const char *what; const int32_t *indptr, *indices; int n_rows; int64_t nnz; int n_cols; };
        const Mat mats[4] = {
            {"m1", m1t ? nullptr : d.m1_indptr, d.m1_indices, a->n_rows_m1, a->nnz_m1, a->n_rows_m2},
            {"m2", m2t ? nullptr : d.m2_indptr, d.m2_indices, a->n_rows_m2, a->nnz_m2, a->n_output_cols},
            {"filter_cols", fm ? d.filter_m_indptr : nullptr, d.filter_m_indices, a->n_rows_m1, a->filter_nnz, a->n_output_cols},
            {"target_cols", tm ? d.target_col_m_indptr : nullptr, d.target_col_m_indices, a->n_rows_m1, a->target_col_nnz, a->n_output_cols}};
        int32_t h[20];
        for (int i = 0; i < 4; ++i) { h[4 * i] = 0; h[4 * i + 1] = 0x7FFFFFFF; h[4 * i + 2] = 0; h[4 * i + 3] = -1; }
        h[16] = h[17] = h[18] = h[19] = 0;       // [16..17] zero count (64 bit), [18] rows with descending ids
        const int32_t *st_c = nullptr;
        TRY(pool.up(h, 20, &st_c));
        int32_t *st = const_cast<int32_t *>(st_c);
        for (int i = 0; i < 4; ++i) {
            if (!mats[i].indptr) continue;
            const long long work = std::max<long long>(mats[i].nnz, mats[i].n_rows);
            hipLaunchKernelGGL(sp_check_csr_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(256 * 16, (work + 255) / 256))), dim3(256), 0, nullptr,
                               mats[i].n_rows, (long long)mats[i].nnz, mats[i].indptr, mats[i].indices, st + 4 * i);
        }
        if (a->flags & SP_FLAG_CHECK_ZEROS) {
            if (!m1t && a->nnz_m1 > 0) hipLaunchKernelGGL(sp_zero_count_kernel, dim3(1024), dim3(256), 0, nullptr, (long long)a->nnz_m1, d.m1_data, (unsigned long long *)(st + 16));
            if (!m2t && a->nnz_m2 > 0) hipLaunchKernelGGL(sp_zero_count_kernel, dim3(1024), dim3(256), 0, nullptr, (long long)a->nnz_m2, d.m2_data, (unsigned long long *)(st + 16));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; ++i) {
            if (!mats[i].indptr) continue;
            const int32_t *e = h + 4 * i;
            if (e[0] & 1) return fail(SP_EINVAL, "%s: indptr[0] is not 0", mats[i].what);
            if (e[0] & 2) return fail(SP_EINVAL, "%s: indptr decreases at row %d", mats[i].what, e[1]);
            if (e[0] & 4) return fail(SP_EINVAL, "%s: indptr[%d] differs from nnz = %lld", mats[i].what, mats[i].n_rows, (long long)mats[i].nnz);
            if (e[2] < 0 || e[3] >= mats[i].n_cols) return fail(SP_EINVAL, "%s: column index out of range [0,%d) (min %d, max %d)", mats[i].what, mats[i].n_cols, e[2], e[3]);
        }
        if ((m1t || (!m2t && (a->flags & SP_FLAG_CHECK_SORTED))) && a->nnz_m2 > 1) {
            // (only now: this kernel walks the rows of m2, whose row pointers have just been validated)
            hipLaunchKernelGGL(sp_rows_sorted_kernel, dim3(std::max(1, std::min(256 * 16, (a->n_rows_m2 + 3) / 4))), dim3(256), 0, nullptr, a->n_rows_m2, d.m2_indptr, d.m2_indices, (unsigned int *)(st + 18));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(h + 18, st + 18, sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        unsigned long long zeros = 0;
        memcpy(&zeros, h + 16, sizeof(zeros));
        if (a->flags & SP_FLAG_CHECK_ZEROS) {
            a->explicit_zeros = (int64_t)zeros;
            if (zeros) return fail(SP_EZEROS, "%llu stored entries are zero: eliminate them first (s_plus.pyx:210-211)", zeros);
        }
        if (h[18]) return fail(SP_EUNSORTED, "%s: %d rows of m2 do not have ascending column ids", m1t ? "SP_FLAG_M1_IS_M2_T" : "SP_FLAG_CHECK_SORTED", h[18]);
        // MATRIX selectors: the kernels look a candidate up in the selector's row by binary search (range_has) — a row whose ids descend
        // would let filtered columns through.  Looked at where the rows are (one wave per row), not trusted from a host-side flag.
        for (int i = 2; i < 4; ++i) {
            if (!mats[i].indptr || mats[i].nnz < 2) continue;
            HIP_TRY(hipMemsetAsync(st + 19, 0, sizeof(int32_t), nullptr));
            hipLaunchKernelGGL(sp_rows_sorted_kernel, dim3(std::max(1, std::min(256 * 16, (mats[i].n_rows + 3) / 4))), dim3(256), 0, nullptr, mats[i].n_rows, mats[i].indptr, mats[i].indices, (unsigned int *)(st + 19));
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(h + 19, st + 19, sizeof(int32_t), hipMemcpyDeviceToHost));
            if (h[19]) return fail(SP_EUNSORTED_SELECTOR, "MATRIX selector %s: %d rows do not have ascending column ids", mats[i].what, h[19]);
        }
    }

    if (a->flags & SP_FLAG_BINARY) {
        // binary=True: ones in the uploaded copies (the zero count above has seen the caller's values, s_plus.pyx:210-217)
        if (!m1t && a->nnz_m1 > 0) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d.m1_data, 0x3F800000, (size_t)a->nnz_m1, nullptr));
        if (!m2t && a->nnz_m2 > 0) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d.m2_data, 0x3F800000, (size_t)a->nnz_m2, nullptr));
        d.flags &= ~SP_FLAG_BINARY;
    }

    d.flags &= ~SP_FLAG_CHECK_SORTED;
    if ((a->flags & SP_FLAG_NORMS_ON_DEVICE) && !m2t && !m1t) {
        // explicit m2: _build_squared_norms (s_plus_utils.pyx:169-201) = row sums of m1^2 (np.add.reduceat's order) and column sums of m2^2
        // (np.bincount's float64 accumulator), then _build_cosine_normalization (:204-228) — from the copies that are here already
        if (a->l1 != 0.f || a->l2 != 0.f) {
            float *sq1 = nullptr, *sq2 = nullptr;
            TRY(pool.alloc((size_t)a->n_rows_m1, &sq1));
            TRY(pool.alloc((size_t)a->n_output_cols, &sq2));
            if (a->n_rows_m1 > 0) {
                sp_csr_sqsums_args q;
                memset(&q, 0, sizeof(q));
                q.struct_size = sizeof(q); q.on_device = 1; q.device = a->device;
                q.n_rows = a->n_rows_m1; q.nnz = a->nnz_m1; q.data = d.m1_data; q.indptr = d.m1_indptr; q.out_rows = sq1;
                TRY(sp_csr_row_sqsums_f32(&q));
            }
            if (a->n_output_cols > 0) {
                sp_csr_colsums_args q;
                memset(&q, 0, sizeof(q));
                q.struct_size = sizeof(q); q.on_device = 1; q.device = a->device;
                q.n_cols = a->n_output_cols; q.square = 1; q.nnz = a->nnz_m2; q.data = d.m2_data; q.indices = d.m2_indices; q.out = sq2;
                TRY(sp_csr_col_sums_f32(&q));
            }
            if (a->l1 != 0.f) { d.Xtversky = sq1; d.Ytversky = sq2; }
            if (a->l2 != 0.f) {
                float *xc = nullptr, *yc = nullptr;
                TRY(pool.alloc((size_t)a->n_rows_m1, &xc));
                TRY(pool.alloc((size_t)a->n_output_cols, &yc));
                if (a->n_rows_m1 > 0) hipLaunchKernelGGL(sp_add_pow_f32_kernel, dim3((a->n_rows_m1 + 255) / 256), dim3(256), 0, nullptr, a->n_rows_m1, (const float *)sq1, xc, a->norm_add, (double)a->norm_c1);
                if (a->n_output_cols > 0) hipLaunchKernelGGL(sp_add_pow_f32_kernel, dim3((a->n_output_cols + 255) / 256), dim3(256), 0, nullptr, a->n_output_cols, (const float *)sq2, yc, a->norm_add, (double)a->norm_c2);
                HIP_TRY(hipGetLastError());
                d.Xcosine = xc; d.Ycosine = yc;
            }
        }
        d.flags &= ~SP_FLAG_NORMS_ON_DEVICE;
    }

    if (a->col_keep && !m2t && a->nnz_m2 > 0) {
        // ARRAY selectors on an explicit m2: the uploaded copy is compacted here (only now: its row pointers and column ids have
        // just been validated)
        const unsigned char *keep = nullptr;
        int *n_indptr = nullptr, *n_idx = nullptr;
        float *n_val = nullptr;
        long long *kept = nullptr, *scan_part = nullptr;
        TRY(pool.up(a->col_keep, (size_t)a->n_output_cols, &keep));
        TRY(pool.alloc((size_t)a->n_rows_m2 + 1, &n_indptr));
        TRY(pool.alloc((size_t)a->nnz_m2, &n_idx));
        TRY(pool.alloc((size_t)a->nnz_m2, &n_val));
        TRY(pool.alloc(1, &kept));
        TRY(pool.alloc((size_t)SCAN_CHUNKS, &scan_part));
        HIP_TRY(hipMemsetAsync(n_indptr, 0, ((size_t)a->n_rows_m2 + 1) * 4, nullptr));
        const int wb = std::max(1, std::min(256 * 16, (a->n_rows_m2 + 3) / 4));
        hipLaunchKernelGGL(sp_keep_count_kernel, dim3(wb), dim3(256), 0, nullptr, a->n_rows_m2, d.m2_indptr, d.m2_indices, keep, n_indptr);
        scan_i32<true>((long long)a->n_rows_m2 + 1, n_indptr, n_indptr, nullptr, kept, scan_part, nullptr);
        hipLaunchKernelGGL(sp_keep_compact_kernel, dim3(wb), dim3(256), 0, nullptr, a->n_rows_m2, d.m2_indptr, d.m2_indices, d.m2_data, keep, n_indptr, n_idx, n_val);
        HIP_TRY(hipGetLastError());
        long long n_kept = 0;
        HIP_TRY(hipMemcpy(&n_kept, kept, sizeof(n_kept), hipMemcpyDeviceToHost));
        d.m2_indptr = n_indptr; d.m2_indices = n_idx; d.m2_data = n_val;
        d.nnz_m2 = n_kept;
    }
    d.col_keep = m2t ? d.col_keep : nullptr;

    const bool csr_out = (a->flags & SP_FLAG_CSR_OUT) != 0;
    bool targets_ascend = true;      // strictly increasing targets: the slots already are in row order
    if (csr_out) {
        for (size_t i = 1; i < nt && targets_ascend; ++i) targets_ascend = a->targets[i] > a->targets[i - 1];
        if (nt * k > 0x7FFFFFFFull) return fail(SP_EINVAL, "SP_FLAG_CSR_OUT: n_targets * k = %zu does not fit int32 row pointers", nt * k);
        d.flags |= SP_FLAG_NO_ROWS_OUT;
    }
    d.flags &= ~(SP_FLAG_CSR_OUT | SP_FLAG_CHECK_ZEROS);
    // the row ids never travel: host threads write them while the device works, the padding of short slots is zeroed afterwards
    const bool want_rows = !(d.flags & SP_FLAG_NO_ROWS_OUT) && a->rows != nullptr;
    d.flags |= SP_FLAG_NO_ROWS_OUT;
    d.rows = nullptr;
    TRY(pool.alloc(nt * k, &d.cols));
    TRY(pool.alloc(nt * k, &d.values));
    d.out_counts = nullptr;
    if (a->out_counts || csr_out || want_rows) TRY(pool.alloc(nt, &d.out_counts));
    {
        // the kernel's workspace comes from the cache as well
        const int64_t need = sp_knn_workspace_bytes(&d);
        if (need < 0) return (int)need;
        unsigned char *w = nullptr;
        TRY(pool.alloc((size_t)need, &w));
        d.workspace = w;
        d.workspace_bytes = need;
    }

    trace.mark("checks, output buffers");
    // While the device works the host is idle: helper threads touch the pages of the caller's (typically fresh, never touched)
    // output arrays so that the copies back do not pay for the page faults.  Output-only memory: writing zeros is harmless.
    HostPrefault prefault;
    if (want_rows) prefault.fill_rows(a->rows, a->targets, nt, k);
    // (SP_FLAG_CSR_OUT with a MATRIX target selector and STRICTLY ASCENDING targets: a row keeps at most the columns its list names and
    // is asked for once — the result has at most target_col_nnz entries, and only that much of cols / values is ever written: see
    // sp_knn.h.  A target that repeats emits its row once per repeat ([7, 7, 7] against a list of 5 columns in row 7 is 15
    // entries): such calls keep the full n_targets * k capacity)
    const size_t out_entries = (csr_out && tm && targets_ascend) ? std::min(nt * k, (size_t)std::max<int64_t>(0, a->target_col_nnz)) : nt * k;
    if (out_entries >= (size_t)1 << 22) {
        prefault.add(a->cols, out_entries * sizeof(int32_t));
        prefault.add(a->values, out_entries * sizeof(float));
        prefault.start();
    }
    // Large results leave in CHUNKS: the target list is cut into four sub-launches (the passes over m2 run once), and while chunk j + 1
    // computes, chunk j is assembled (CSR: its slots' non-zeros compacted; strictly increasing targets make slot order row order) and
    // copied to the host on a second stream — of the ~16 ms that assembly + 0.8 GB of PCIe cost at the C2 size only the last chunk's
    // share stays exposed (44 % of the public call was transfers and glue, serial with the kernel).
    // (SIMILARIPY_AMD_CHUNK_MIN_ENTRIES: the threshold in output entries, for tests at small sizes; SIMILARIPY_AMD_NO_CHUNKS: off)
    const char *cmin_env = getenv("SIMILARIPY_AMD_CHUNK_MIN_ENTRIES");
    const size_t chunk_min = cmin_env ? (size_t)strtoull(cmin_env, nullptr, 10) : ((size_t)1 << 24);
    const bool chunked = !(d.flags & SP_FLAG_TIME_KERNEL) && nt * k >= chunk_min && nt >= 64 && (!csr_out || targets_ascend) &&
                         getenv("SIMILARIPY_AMD_NO_CHUNKS") == nullptr;
    ChunkHook hook;
    struct ChunkSync {
        hipStream_t s2 = nullptr;
        std::vector<hipEvent_t> ev;
        ~ChunkSync() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); if (s2) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); } }
    } cs_;
    if (chunked) {
        hook.n_chunks = 4;
        for (int j = 0; j <= hook.n_chunks; ++j) hook.bounds.push_back(nt * (size_t)j / (size_t)hook.n_chunks);
        HIP_TRY(hipStreamCreateWithFlags(&cs_.s2, hipStreamNonBlocking));
        cs_.ev.resize((size_t)hook.n_chunks, nullptr);
        for (auto &e : cs_.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hook.after_launch = [&](int j) -> int { HIP_TRY(hipEventRecord(cs_.ev[(size_t)j], nullptr)); return SP_OK; };
    }
    g_p3_zero_counter = nullptr;
    int rc = run_device(&d, chunked ? &hook : nullptr);
    if (rc) return rc;
    // SP_FLAG_P3_PREP: entries that underflowed to 0.0 in the divide or the power (checked once the device work has been waited for)
    auto p3_underflow = [&]() -> int {
        if (!g_p3_zero_counter) return SP_OK;
        unsigned long long z = 0;
        HIP_TRY(hipMemcpy(&z, g_p3_zero_counter, sizeof(z), hipMemcpyDeviceToHost));
        g_p3_zero_counter = nullptr;
        if (z) {
            a->explicit_zeros = (int64_t)z;
            return fail(SP_EUNDERFLOW, "SP_FLAG_P3_PREP: %llu stored entries became 0.0 when they were L1-normalised and raised to %g; the reference drops "
                        "them before its kernel runs (similarity.py:410-415, then s_plus.pyx:210-211): preprocess on the host and call again", z, (double)a->p3_alpha);
        }
        return SP_OK;
    };
    if (chunked) {
        // (every chunk's launches are queued; the host now follows them chunk by chunk on the second stream)
        int *slot_nnz = nullptr, *slot_off = nullptr, *o_idx = nullptr;
        float *o_val = nullptr;
        long long *totals = nullptr, *scan_part = nullptr;
        if (csr_out) {
            TRY(pool.alloc(nt, &slot_nnz));
            TRY(pool.alloc(nt + (size_t)hook.n_chunks, &slot_off));
            TRY(pool.alloc(nt * k, &o_idx));
            TRY(pool.alloc(nt * k, &o_val));
            TRY(pool.alloc((size_t)hook.n_chunks, &totals));
            TRY(pool.alloc((size_t)SCAN_CHUNKS, &scan_part));
        }
        size_t running = 0;
        const bool progress = (a->flags & SP_FLAG_PROGRESS) != 0;
        for (int j = 0; j < hook.n_chunks; ++j) {
            const size_t s0 = hook.bounds[(size_t)j], s1 = hook.bounds[(size_t)j + 1], ns = s1 - s0;
            HIP_TRY(hipStreamWaitEvent(cs_.s2, cs_.ev[(size_t)j], 0));
            if (!ns) continue;
            if (csr_out) {
                const int wb = (int)std::max<size_t>(1, std::min<size_t>(256 * 16, (ns + 3) / 4));
                int *off_j = slot_off + s0 + (size_t)j;                      // (ns + 1 entries)
                hipLaunchKernelGGL(sp_chunk_slot_nnz_kernel, dim3(wb), dim3(256), 0, cs_.s2, (int)ns, (int)k, d.out_counts + s0, d.values + s0 * k, slot_nnz + s0);
                scan_i32<false>((long long)ns, slot_nnz + s0, off_j, nullptr, totals + j, scan_part, cs_.s2);
                hipLaunchKernelGGL(sp_chunk_compact_kernel, dim3(wb), dim3(256), 0, cs_.s2, (int)ns, (int)k, d.out_counts + s0, d.cols + s0 * k, d.values + s0 * k,
                                   (const int *)off_j, o_idx + s0 * k, o_val + s0 * k);
                HIP_TRY(hipGetLastError());
                long long nnz_j = 0;
                HIP_TRY(hipMemcpyAsync(&nnz_j, totals + j, sizeof(nnz_j), hipMemcpyDeviceToHost, cs_.s2));
                HIP_TRY(hipStreamSynchronize(cs_.s2));
                if (j == 0) prefault.join();
                if (running + (size_t)nnz_j > out_entries) return fail(SP_EINVAL, "internal: the CSR result (%zu entries so far) exceeds the documented capacity of cols / values (%zu)", running + (size_t)nnz_j, out_entries);
                if (nnz_j > 0) {
                    HIP_TRY(hipMemcpyAsync(a->cols + running, o_idx + s0 * k, (size_t)nnz_j * 4, hipMemcpyDeviceToHost, cs_.s2));
                    HIP_TRY(hipMemcpyAsync(a->values + running, o_val + s0 * k, (size_t)nnz_j * 4, hipMemcpyDeviceToHost, cs_.s2));
                }
                running += (size_t)nnz_j;
                if (progress) { HIP_TRY(hipStreamSynchronize(cs_.s2)); fprintf(stderr, "[similaripy_amd] rows done: %zu / %zu\n", s1, nt); }
            } else {
                if (j == 0) { HIP_TRY(hipStreamSynchronize(cs_.s2)); prefault.join(); }
                HIP_TRY(hipMemcpyAsync(a->cols + s0 * k, d.cols + s0 * k, ns * k * sizeof(int32_t), hipMemcpyDeviceToHost, cs_.s2));
                HIP_TRY(hipMemcpyAsync(a->values + s0 * k, d.values + s0 * k, ns * k * sizeof(float), hipMemcpyDeviceToHost, cs_.s2));
                if (progress) { HIP_TRY(hipStreamSynchronize(cs_.s2)); fprintf(stderr, "[similaripy_amd] rows done: %zu / %zu\n", s1, nt); }
            }
        }
        if (csr_out) {
            // the row pointers: one count over all slots + one scan (the entries are on their way already, in row order)
            const int n_rows = a->n_rows_m1;
            int *indptr = nullptr;
            long long *total = nullptr;
            TRY(pool.alloc((size_t)n_rows + 1, &indptr));
            TRY(pool.alloc(1, &total));
            HIP_TRY(hipMemsetAsync(indptr, 0, ((size_t)n_rows + 1) * 4, cs_.s2));
            const int wb = (int)std::max<size_t>(1, std::min<size_t>(256 * 16, (nt + 3) / 4));
            hipLaunchKernelGGL(sp_slot_nnz_kernel, dim3(wb), dim3(256), 0, cs_.s2, (int)nt, (int)k, d.targets, d.out_counts, d.values, indptr);
            scan_i32<true>((long long)n_rows + 1, indptr, indptr, nullptr, total, scan_part, cs_.s2);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(a->csr_indptr, indptr, ((size_t)n_rows + 1) * 4, hipMemcpyDeviceToHost, cs_.s2));
            a->csr_nnz = (int64_t)running;
        }
        if (a->out_counts) HIP_TRY(hipMemcpyAsync(a->out_counts, d.out_counts, nt * sizeof(int32_t), hipMemcpyDeviceToHost, cs_.s2));
        HIP_TRY(hipStreamSynchronize(cs_.s2));
        trace.mark("row kernels, chunked assembly + result to the host");
        TRY(p3_underflow());
        if (want_rows) {
            prefault.join();
            std::vector<int32_t> cnt_tmp;
            const int32_t *cnt = a->out_counts;
            if (!cnt) {
                cnt_tmp.resize(nt);
                HIP_TRY(hipMemcpy(cnt_tmp.data(), d.out_counts, nt * sizeof(int32_t), hipMemcpyDeviceToHost));
                cnt = cnt_tmp.data();
            }
            for (size_t i = 0; i < nt; ++i)
                if ((size_t)cnt[i] < k) memset(a->rows + i * k + cnt[i], 0, (k - (size_t)cnt[i]) * sizeof(int32_t));
        }
        a->kernel_ms = d.kernel_ms;
        a->passes_total = d.passes_total;
        a->num_wgs_used = d.num_wgs_used;
        memcpy(a->phase_cycles, d.phase_cycles, sizeof(a->phase_cycles));
        a->reserved[1] = d.reserved[1]; a->reserved[2] = d.reserved[2]; a->reserved[3] = d.reserved[3];
        return SP_OK;
    }
    trace.mark("transpose, norms, row kernels");
    prefault.join();
    trace.mark("output pages touched (host)");
    if (csr_out) {
        // counting sort of the slots by row (coo_to_csr.h:28-71) with the zeros left out (s_plus.pyx:424): targets ascend, so
        // the slots already are in row order — per-slot non-zero counts, a scan, one compaction pass, and only the CSR travels
        const int n_rows = a->n_rows_m1;
        int *indptr = nullptr, *o_idx = nullptr;
        float *o_val = nullptr;
        long long *total = nullptr;
        TRY(pool.alloc((size_t)n_rows + 1, &indptr));
        TRY(pool.alloc(nt * k, &o_idx));
        TRY(pool.alloc(nt * k, &o_val));
        long long *scan_part = nullptr;
        TRY(pool.alloc(1, &total));
        TRY(pool.alloc((size_t)SCAN_CHUNKS, &scan_part));
        HIP_TRY(hipMemsetAsync(indptr, 0, ((size_t)n_rows + 1) * 4, nullptr));
        const int wb = (int)std::max<size_t>(1, std::min<size_t>(256 * 16, (nt + 3) / 4));
        if (targets_ascend) {
            hipLaunchKernelGGL(sp_slot_nnz_kernel, dim3(wb), dim3(256), 0, nullptr, (int)nt, (int)k, d.targets, d.out_counts, d.values, indptr);
            scan_i32<true>((long long)n_rows + 1, indptr, indptr, nullptr, total, scan_part, nullptr);      // (indptr[0] = 0: in place it becomes the row pointers)
            hipLaunchKernelGGL(sp_csr_compact_kernel, dim3(wb), dim3(256), 0, nullptr, (int)nt, (int)k, d.targets, d.out_counts, d.cols, d.values, indptr, o_idx, o_val);
        } else {
            // any order, repeats included (target_rows=[7, 2, 7]): the stable counting sort by row of coo_to_csr.h:28-71
            int *slot_nnz = nullptr, *slot_off = nullptr, *bstart = nullptr, *cursor = nullptr, *bucket = nullptr;
            long long *total2 = nullptr;
            TRY(pool.alloc(nt, &slot_nnz));
            TRY(pool.alloc(nt, &slot_off));
            TRY(pool.alloc((size_t)n_rows + 1, &bstart));
            TRY(pool.alloc((size_t)n_rows + 1, &cursor));
            TRY(pool.alloc(nt, &bucket));
            TRY(pool.alloc(1, &total2));
            HIP_TRY(hipMemsetAsync(bstart, 0, ((size_t)n_rows + 1) * 4, nullptr));
            HIP_TRY(hipMemsetAsync(cursor, 0, ((size_t)n_rows + 1) * 4, nullptr));
            hipLaunchKernelGGL(sp_slot_nnz_any_kernel, dim3(wb), dim3(256), 0, nullptr, (int)nt, (int)k, d.targets, d.out_counts, d.values, slot_nnz, indptr, bstart);
            scan_i32<true>((long long)n_rows + 1, indptr, indptr, nullptr, total, scan_part, nullptr);
            scan_i32<true>((long long)n_rows + 1, bstart, bstart, nullptr, total2, scan_part, nullptr);
            const int tb = (int)std::max<size_t>(1, std::min<size_t>(256 * 16, (nt + 255) / 256));
            const int rb = (int)std::max<size_t>(1, std::min<size_t>(256 * 16, ((size_t)n_rows + 255) / 256));
            hipLaunchKernelGGL(sp_slot_scatter_kernel, dim3(tb), dim3(256), 0, nullptr, (int)nt, d.targets, bstart, cursor, bucket);
            hipLaunchKernelGGL(sp_slot_offsets_kernel, dim3(rb), dim3(256), 0, nullptr, n_rows, bstart, bucket, slot_nnz, slot_off);
            hipLaunchKernelGGL(sp_csr_compact_any_kernel, dim3(wb), dim3(256), 0, nullptr, (int)nt, (int)k, d.targets, d.out_counts, d.cols, d.values, indptr, slot_off, o_idx, o_val);
        }
        HIP_TRY(hipGetLastError());
        long long nnz = 0;
        HIP_TRY(hipMemcpy(&nnz, total, sizeof(nnz), hipMemcpyDeviceToHost));
        a->csr_nnz = nnz;
        if ((size_t)std::max<long long>(0, nnz) > out_entries) return fail(SP_EINVAL, "internal: the CSR result (%lld entries) exceeds the documented capacity of cols / values (%zu)", nnz, out_entries);
        HIP_TRY(hipMemcpy(a->csr_indptr, indptr, ((size_t)n_rows + 1) * 4, hipMemcpyDeviceToHost));
        if (nnz > 0) {
            HIP_TRY(hipMemcpy(a->cols, o_idx, (size_t)nnz * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(a->values, o_val, (size_t)nnz * 4, hipMemcpyDeviceToHost));
        }
        if (a->out_counts) HIP_TRY(hipMemcpy(a->out_counts, d.out_counts, nt * sizeof(int32_t), hipMemcpyDeviceToHost));
    } else {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(a->cols, d.cols, nt * k * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(a->values, d.values, nt * k * sizeof(float), hipMemcpyDeviceToHost));
        if (a->out_counts) HIP_TRY(hipMemcpy(a->out_counts, d.out_counts, nt * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (want_rows) {
            // padding is (0, 0, 0.0) (s_plus.h:246-262 leaves the calloc'ed tail untouched): zero the row ids behind every short slot
            std::vector<int32_t> cnt_tmp;
            const int32_t *cnt = a->out_counts;
            if (!cnt) {
                cnt_tmp.resize(nt);
                HIP_TRY(hipMemcpy(cnt_tmp.data(), d.out_counts, nt * sizeof(int32_t), hipMemcpyDeviceToHost));
                cnt = cnt_tmp.data();
            }
            for (size_t i = 0; i < nt; ++i)
                if ((size_t)cnt[i] < k) memset(a->rows + i * k + cnt[i], 0, (k - (size_t)cnt[i]) * sizeof(int32_t));
        }
    }
    trace.mark("assembly, result to the host");
    if (a->flags & SP_FLAG_PROGRESS) fprintf(stderr, "[similaripy_amd] rows done: %zu / %zu\n", nt, nt);
    TRY(p3_underflow());
    a->kernel_ms = d.kernel_ms;
    a->passes_total = d.passes_total;
    a->num_wgs_used = d.num_wgs_used;
    memcpy(a->phase_cycles, d.phase_cycles, sizeof(a->phase_cycles));
    a->reserved[1] = d.reserved[1]; a->reserved[2] = d.reserved[2]; a->reserved[3] = d.reserved[3];
    return SP_OK;
}

// ---------------------------------------------------------------------------------------------
// ABI 5: one host-mode call over several devices (sp_knn_args::n_devices / device_ids)
// ---------------------------------------------------------------------------------------------
template <typename F>
void parallel_ranges(size_t n, size_t min_chunk, F &&f) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_thr = std::max<size_t>(1, std::min<size_t>({(size_t)32, hw ? (size_t)hw : (size_t)1, (n + min_chunk - 1) / std::max<size_t>(1, min_chunk)}));
    if (n_thr <= 1) { f((size_t)0, n); return; }
    const size_t per = (n + n_thr - 1) / n_thr;
    std::vector<std::thread> th;
    for (size_t lo = 0; lo < n; lo += per) {
        const size_t hi = std::min(n, lo + per);
        try { th.emplace_back([&f, lo, hi]() { f(lo, hi); }); } catch (...) { f(lo, hi); }
    }
    for (auto &t : th) t.join();
}

// THE partition cost model (one place: the in-library "threads" route below, and — through sp_knn_target_costs / sp_knn_partition — the
// one-process-per-GPU route of similaripy_amd/distributed.py; VERDICT r5 #6: two copies had diverged).  cost[i] of target slot i, in MAC
// equivalents:
//     MACs(targets[i])
//   + a fixed toll per row: 30 k for a row of the sparse kernels (queue, setup, bitmap clear, selection, write-out whatever its length:
//     profiles/r03_c2_phases.txt), 3 per output column for a row of the generic kernel, which walks every column window whatever the row
//     holds (SIMILARIPY_AMD_GENERIC_TOLL_PER_COL; profiles/r04_exp_strong_scaling_c4.txt)
//   + for a HEAVY generic row — one the launch cuts into column-window pieces: MACs >= 2 x split_piece_macs, the launch's own rule — a
//     price per m1 ENTRY (SIMILARIPY_AMD_HEAVY_ENTRY_MACS, default 2 100: every fine window of such a row walks all of its segments for a
//     handful of elements; least squares over the slices of N = 1 .. 8 at the MovieLens-32M shape, profiles/r05_exp_dropped.txt).
// Which rows are "sparse" restates sp_row_desc_kernel's rule from sizes.
constexpr double ROW_TOLL_MACS = 30000.0;
int target_costs(const sp_knn_args *a, std::vector<double> *cost) {
    const size_t nt = (size_t)a->n_targets;
    cost->assign(nt, ROW_TOLL_MACS);
    const bool m2t = (a->flags & SP_FLAG_M2_IS_M1_T) != 0, m1t = (a->flags & SP_FLAG_M1_IS_M2_T) != 0;
    const double n_cols_d = (double)std::max(1, a->n_output_cols);
    double toll_per_col = 3.0, heavy_entry = 2100.0;
    if (const char *e = getenv("SIMILARIPY_AMD_GENERIC_TOLL_PER_COL")) { const double v = atof(e); if (v > 0.0) toll_per_col = v; }
    if (const char *e = getenv("SIMILARIPY_AMD_HEAVY_ENTRY_MACS")) { const double v = atof(e); if (v >= 0.0) heavy_entry = v; }
    // the piece size the launch will use for these sizes (the persistent generic workgroups of a 256-CU device when none is visible)
    double heavy_from = 1e300;
    {
        sp_knn_args b = *a;
        if (m2t) { b.nnz_m2 = a->nnz_m1; }
        if (m1t) { b.nnz_m1 = a->nnz_m2; }
        int n_cus = 256;
        if (sp_device_count() > 0) {
            hipDeviceProp_t prop;
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cus = prop.multiProcessorCount;
        }
        Config c{};
        if (make_config(&b, n_cus, &c) == SP_OK && c.split_pmax >= 2 && c.n_splits >= 1) heavy_from = 2.0 * (double)split_piece_macs(&b, c.wgs_generic);
    }
    auto priced = [&](double m, long long nnz1) {
        const bool sparse_row = 0.5 * m * m / n_cols_d <= 0.30 * 4096.0 && nnz1 <= 256 && a->n_output_cols > 16384;
        const bool heavy = !sparse_row && m >= heavy_from;
        return m + (sparse_row ? ROW_TOLL_MACS : toll_per_col * n_cols_d) + (heavy ? heavy_entry * (double)nnz1 : 0.0);
    };
    if (m1t) {
        // m1 = m2^T does not exist on the host: MACs(t) = sum over the entries (u, t) of m2 of len(m2 row u), scattered by column;
        // nnz1(t) = the number of such entries (the same pass).  Priced like every other row (ratings-shaped data — rows of
        // the generic kernel — were priced without its toll here, the default route of the public item-item call)
        std::vector<double> macs((size_t)a->n_rows_m1, 0.0);
        std::vector<int> nnz1((size_t)a->n_rows_m1, 0);
        for (int u = 0; u < a->n_rows_m2; ++u) {
            const int lo = std::max(0, a->m2_indptr[u]), hi = (int)std::min<int64_t>(a->nnz_m2, a->m2_indptr[u + 1]);
            const double len = (double)std::max(0, hi - lo);
            for (int p = lo; p < hi; ++p) {
                const int t = a->m2_indices[p];
                if (t >= 0 && t < a->n_rows_m1) { macs[(size_t)t] += len; ++nnz1[(size_t)t]; }
            }
        }
        for (size_t i = 0; i < nt; ++i) (*cost)[i] = priced(macs[(size_t)a->targets[i]], nnz1[(size_t)a->targets[i]]);
        return SP_OK;
    }
    std::vector<int> len2((size_t)a->n_rows_m2, 0);
    if (m2t) {
        // m2 = m1^T does not exist on the host: the length of its row u is the number of m1 entries in column u
        const size_t nnz = (size_t)a->nnz_m1;
        std::vector<std::vector<int>> part;
        std::mutex mu;
        parallel_ranges(nnz, (size_t)1 << 22, [&](size_t lo, size_t hi) {
            std::vector<int> loc((size_t)a->n_rows_m2, 0);
            for (size_t p = lo; p < hi; ++p) {
                const int u = a->m1_indices[p];
                if (u >= 0 && u < a->n_rows_m2) ++loc[(size_t)u];
            }
            std::lock_guard<std::mutex> lk(mu);
            part.push_back(std::move(loc));
        });
        for (auto &v : part) for (size_t u = 0; u < v.size(); ++u) len2[u] += v[u];
    } else {
        for (int u = 0; u < a->n_rows_m2; ++u) len2[(size_t)u] = a->m2_indptr[u + 1] - a->m2_indptr[u];
    }
    parallel_ranges(nt, (size_t)1 << 16, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const int t = a->targets[i];
            double m = 0.0;
            // (the arrays are validated on the device later: a malformed row pointer must not take the host down here)
            const int p_lo = std::max(0, a->m1_indptr[t]), p_hi = (int)std::min<int64_t>(a->nnz_m1, a->m1_indptr[t + 1]);
            for (int p = p_lo; p < p_hi; ++p) {
                const int u = a->m1_indices[p];
                if (u >= 0 && u < a->n_rows_m2) m += (double)len2[(size_t)u];
            }
            (*cost)[i] = priced(m, (long long)(p_hi - p_lo));
        }
    });
    return SP_OK;
}

// bounds[0 .. n_parts] of contiguous slices of equal cumulative cost: slice r starts behind the first slot at which the running cost
// reaches r / n_parts of the total
void partition_by_cost(const std::vector<double> &cost, int n_parts, std::vector<size_t> *bounds) {
    const size_t nt = cost.size();
    bounds->assign((size_t)n_parts + 1, 0);
    double total = 0.0;
    for (double c : cost) total += c;
    double run = 0.0;
    size_t i = 0;
    for (int r = 1; r < n_parts; ++r) {
        const double want = total * (double)r / (double)n_parts;
        while (i < nt && run < want) run += cost[i++];
        (*bounds)[(size_t)r] = i;
    }
    (*bounds)[(size_t)n_parts] = nt;
}

int check_cost_args(const sp_knn_args *a) {
    if (!a) return fail(SP_EINVAL, "args is NULL");
    if (a->struct_size != sizeof(sp_knn_args)) return fail(SP_EINVAL, "sp_knn_args size mismatch: caller %u, library %zu", a->struct_size, sizeof(sp_knn_args));
    if (a->on_device) return fail(SP_EINVAL, "the cost model reads the CSR structure on the host (on_device must be 0)");
    if (a->n_targets < 0 || a->n_rows_m1 < 0 || a->n_rows_m2 < 0) return fail(SP_EINVAL, "negative dimension");
    const bool m2t = (a->flags & SP_FLAG_M2_IS_M1_T) != 0, m1t = (a->flags & SP_FLAG_M1_IS_M2_T) != 0;
    if (a->n_targets > 0 && !a->targets) return fail(SP_EINVAL, "targets is NULL");
    if (!m1t && (!a->m1_indptr || (a->nnz_m1 > 0 && !a->m1_indices))) return fail(SP_EINVAL, "m1 structure pointers are NULL");
    if (!m2t && !a->m2_indptr) return fail(SP_EINVAL, "m2_indptr is NULL");
    if (m1t && a->nnz_m2 > 0 && !a->m2_indices) return fail(SP_EINVAL, "m2_indices is NULL");
    for (int i = 0; i < a->n_targets; ++i)
        if (a->targets[i] < 0 || a->targets[i] >= a->n_rows_m1) return fail(SP_EINVAL, "targets[%d]=%d out of range [0,%d)", i, a->targets[i], a->n_rows_m1);
    return SP_OK;
}

int run_host_multi(sp_knn_args *a) {
    const int nd = a->n_devices;
    const size_t nt = (size_t)a->n_targets, k = (size_t)a->k;
    const int ndev = sp_device_count();
    std::vector<int> devs((size_t)nd);
    for (int i = 0; i < nd; ++i) {
        devs[(size_t)i] = a->device_ids ? a->device_ids[i] : i;
        if (devs[(size_t)i] < 0 || devs[(size_t)i] >= ndev) return fail(SP_EINVAL, "device_ids[%d] = %d out of range (have %d)", i, devs[(size_t)i], ndev);
        // (SIMILARIPY_AMD_ALLOW_REPEATED_DEVICES: the sharding, the per-device threads and the joins of the pieces on a one-GPU box — tests)
        for (int j = 0; j < i && getenv("SIMILARIPY_AMD_ALLOW_REPEATED_DEVICES") == nullptr; ++j)
            if (devs[(size_t)j] == devs[(size_t)i]) return fail(SP_EINVAL, "device_ids holds device %d twice", devs[(size_t)i]);
    }
    for (size_t i = 0; i < nt; ++i)
        if (a->targets[i] < 0 || a->targets[i] >= a->n_rows_m1)
            return fail(SP_EINVAL, "targets[%zu]=%d out of range [0,%d)", i, a->targets[i], a->n_rows_m1);
    const bool csr_out = (a->flags & SP_FLAG_CSR_OUT) != 0;
    if (csr_out)
        for (size_t i = 1; i < nt; ++i)
            if (a->targets[i] <= a->targets[i - 1])
                return fail(SP_EINVAL, "SP_FLAG_CSR_OUT over several devices needs strictly increasing targets (targets[%zu] = %d follows %d)", i, a->targets[i], a->targets[i - 1]);
    // contiguous slices of equal cumulative cost
    std::vector<double> cost;
    TRY(target_costs(a, &cost));
    std::vector<size_t> bounds;
    partition_by_cost(cost, nd, &bounds);
    struct Part { sp_knn_args args; int rc = SP_OK; std::string err; std::vector<int32_t> indptr; };
    std::vector<Part> parts((size_t)nd);
    std::vector<std::thread> th;
    for (int r = 0; r < nd; ++r) {
        Part &P = parts[(size_t)r];
        const size_t lo = bounds[(size_t)r], hi = bounds[(size_t)r + 1];
        P.args = *a;
        P.args.n_devices = 0; P.args.device_ids = nullptr;
        P.args.device = devs[(size_t)r];
        P.args.n_targets = (int32_t)(hi - lo);
        P.args.targets = a->targets + lo;
        if (a->rows) P.args.rows = a->rows + lo * k;
        P.args.cols = a->cols + lo * k;
        P.args.values = a->values + lo * k;
        if (a->out_counts) P.args.out_counts = a->out_counts + lo;
        if (csr_out) { P.indptr.assign((size_t)a->n_rows_m1 + 1, 0); P.args.csr_indptr = P.indptr.data(); P.args.csr_nnz = 0; }
        if (hi == lo) continue;
        th.emplace_back([&P]() {
            P.rc = run_host(&P.args);
            if (P.rc) P.err = g_err;            // (g_err is the worker thread's)
        });
    }
    for (auto &t : th) t.join();
    for (int r = 0; r < nd; ++r) {
        Part &P = parts[(size_t)r];
        if (P.rc) {
            a->explicit_zeros = P.args.explicit_zeros;
            return fail(P.rc, "device %d: %s", devs[(size_t)r], P.err.c_str());
        }
    }
    a->kernel_ms = 0.f; a->passes_total = 0; a->num_wgs_used = 0; a->explicit_zeros = 0;
    memset(a->phase_cycles, 0, sizeof(a->phase_cycles));
    a->reserved[1] = a->reserved[2] = a->reserved[3] = 0;
    for (int r = 0; r < nd; ++r) {
        const Part &P = parts[(size_t)r];
        if (bounds[(size_t)r + 1] == bounds[(size_t)r]) continue;
        a->kernel_ms = std::max(a->kernel_ms, P.args.kernel_ms);                       // the devices run side by side
        a->passes_total += P.args.passes_total;
        a->num_wgs_used += P.args.num_wgs_used;
        for (int i = 0; i < 12; ++i) a->phase_cycles[i] += P.args.phase_cycles[i];
        for (int i = 1; i <= 3; ++i) a->reserved[i] = std::max(a->reserved[i], P.args.reserved[i]);
    }
    if (csr_out) {
        // the targets ascend and the slices are contiguous: device r's entries follow device r-1's, and the row pointers add up
        // (every piece's indptr counts that piece's entries in the rows below i)
        int64_t total = 0;
        for (int r = 0; r < nd; ++r) {
            Part &P = parts[(size_t)r];
            const size_t lo = bounds[(size_t)r], hi = bounds[(size_t)r + 1];
            if (hi == lo) continue;
            const int64_t n = P.args.csr_nnz;
            if (n > 0 && (size_t)total != lo * k) {
                memmove(a->cols + total, a->cols + lo * k, (size_t)n * sizeof(int32_t));
                memmove(a->values + total, a->values + lo * k, (size_t)n * sizeof(float));
            }
            total += n;
        }
        if (total > 0x7FFFFFFFLL) return fail(SP_EINVAL, "SP_FLAG_CSR_OUT: %lld entries do not fit int32 row pointers", (long long)total);
        parallel_ranges((size_t)a->n_rows_m1 + 1, (size_t)1 << 18, [&](size_t lo_, size_t hi_) {
            for (size_t i = lo_; i < hi_; ++i) {
                int32_t v = 0;
                for (int r = 0; r < nd; ++r) if (!parts[(size_t)r].indptr.empty()) v += parts[(size_t)r].indptr[i];
                a->csr_indptr[i] = v;
            }
        });
        a->csr_nnz = total;
    }
    return SP_OK;
}

}  // namespace

extern "C" {

int sp_abi_version(void) { return SP_KNN_ABI_VERSION; }

int sp_knn_target_costs(const sp_knn_args *a, double *cost) {
    g_err[0] = 0;
    TRY(check_cost_args(a));
    if (a->n_targets > 0 && !cost) return fail(SP_EINVAL, "cost is NULL");
    std::vector<double> c;
    TRY(target_costs(a, &c));
    for (size_t i = 0; i < c.size(); ++i) cost[i] = c[i];
    return SP_OK;
}

int sp_knn_partition(const sp_knn_args *a, int n_parts, int64_t *bounds) {
    g_err[0] = 0;
    TRY(check_cost_args(a));
    if (n_parts < 1 || !bounds) return fail(SP_EINVAL, "n_parts must be >= 1 and bounds non-NULL");
    std::vector<double> c;
    TRY(target_costs(a, &c));
    std::vector<size_t> b;
    partition_by_cost(c, n_parts, &b);
    for (int r = 0; r <= n_parts; ++r) bounds[r] = (int64_t)b[(size_t)r];
    return SP_OK;
}

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
    if (a->flags & (SP_FLAG_M2_IS_M1_T | SP_FLAG_M1_IS_M2_T)) {
        sp_knn_args plain;
        M2tLayout L;
        rc = m2t_layout(a, n_cus, &plain, &L);
        if (rc) return rc;
        return (int64_t)L.total;
    }
    Config c{};
    rc = make_config(a, n_cus, &c);
    if (rc) return rc;
    return (int64_t)c.ws_total;
}

int64_t sp_csr_transpose_workspace_bytes(const sp_csr_transpose_args *a) {
    if (!a || a->struct_size != sizeof(sp_csr_transpose_args)) return fail(SP_EINVAL, "sp_csr_transpose_args size mismatch");
    if (a->nnz < 0 || a->n_cols < 0) return fail(SP_EINVAL, "negative size");
    return (int64_t)transpose_ws_bytes(a->nnz, a->n_cols);
}

int sp_csr_transpose_f32_i32(sp_csr_transpose_args *a) {
    g_err[0] = 0;
    if (!a || a->struct_size != sizeof(sp_csr_transpose_args)) return fail(SP_EINVAL, "sp_csr_transpose_args size mismatch");
    if (a->n_rows < 0 || a->n_cols < 0 || a->nnz < 0 || a->nnz > 0x7FFFFFFFLL) return fail(SP_EINVAL, "bad shape / nnz");
    if (!a->indptr || !a->out_indptr || (a->nnz > 0 && (!a->data || !a->indices || !a->out_data || !a->out_indices)))
        return fail(SP_EINVAL, "NULL input/output pointer");
    const int ndev = sp_device_count();
    if (ndev <= 0) return fail(SP_ENODEVICE, "no HIP device visible: similaripy_amd has no CPU fallback");
    if (a->device < 0 || a->device >= ndev) return fail(SP_EINVAL, "device %d out of range (have %d)", a->device, ndev);
    HIP_TRY(hipSetDevice(a->device));
    const size_t need = transpose_ws_bytes(a->nnz, a->n_cols);
    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    a->kernel_ms = 0.f;
    if (a->on_device) {
        hipStream_t stream = (hipStream_t)a->stream;
        unsigned char *ws = (unsigned char *)a->workspace;
        bool own = false;
        if (!ws) { HIP_TRY(hipMalloc((void **)&ws, need)); own = true; }
        else if (a->workspace_bytes < (int64_t)need) return fail(SP_EWORKSPACE, "workspace too small: need %zu bytes, got %lld", need, (long long)a->workspace_bytes);
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (timed) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
        int rc = transpose_device(a->n_rows, a->n_cols, a->nnz, a->data, a->indices, a->indptr, a->out_data, a->out_indices, a->out_indptr, ws, need, stream);
        if (timed) {
            if (!rc) { HIP_TRY(hipEventRecord(ev1, stream)); HIP_TRY(hipEventSynchronize(ev1)); HIP_TRY(hipEventElapsedTime(&a->kernel_ms, ev0, ev1)); }
            (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
        }
        if (own) { (void)hipStreamSynchronize(stream); (void)hipFree(ws); }
        return rc;
    }
    // host buffers in, host buffers out
    for (int64_t i = 0; i < a->nnz; ++i)
        if (a->indices[i] < 0 || a->indices[i] >= a->n_cols) return fail(SP_EINVAL, "indices[%lld]=%d out of range [0,%d)", (long long)i, a->indices[i], a->n_cols);
    DevPool pool;
    pool.device = a->device;
    const float *d_data; const int32_t *d_indices, *d_indptr;
    float *o_data; int32_t *o_indices, *o_indptr;
    unsigned char *ws;
    TRY(pool.up(a->data, (size_t)a->nnz, &d_data));
    TRY(pool.up(a->indices, (size_t)a->nnz, &d_indices));
    TRY(pool.up(a->indptr, (size_t)a->n_rows + 1, &d_indptr));
    TRY(pool.alloc((size_t)a->nnz, &o_data));
    TRY(pool.alloc((size_t)a->nnz, &o_indices));
    TRY(pool.alloc((size_t)a->n_cols + 1, &o_indptr));
    TRY(pool.alloc(need, &ws));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, nullptr)); }
    int rc = transpose_device(a->n_rows, a->n_cols, a->nnz, d_data, d_indices, d_indptr, o_data, o_indices, o_indptr, ws, need, nullptr);
    if (timed) {
        if (!rc) { HIP_TRY(hipEventRecord(ev1, nullptr)); HIP_TRY(hipEventSynchronize(ev1)); HIP_TRY(hipEventElapsedTime(&a->kernel_ms, ev0, ev1)); }
        (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    }
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (a->nnz > 0) {
        HIP_TRY(hipMemcpy(a->out_data, o_data, (size_t)a->nnz * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(a->out_indices, o_indices, (size_t)a->nnz * 4, hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(a->out_indptr, o_indptr, ((size_t)a->n_cols + 1) * 4, hipMemcpyDeviceToHost));
    return SP_OK;
}

int sp_csr_row_sqsums_f32(sp_csr_sqsums_args *a) {
    g_err[0] = 0;
    if (!a || a->struct_size != sizeof(sp_csr_sqsums_args)) return fail(SP_EINVAL, "sp_csr_sqsums_args size mismatch");
    if (a->n_rows < 0 || a->nnz < 0 || a->nnz > 0x7FFFFFFFLL) return fail(SP_EINVAL, "bad shape / nnz");
    if (!a->indptr || (a->nnz > 0 && !a->data)) return fail(SP_EINVAL, "NULL input pointer");
    const int ndev = sp_device_count();
    if (ndev <= 0) return fail(SP_ENODEVICE, "no HIP device visible: similaripy_amd has no CPU fallback");
    if (a->device < 0 || a->device >= ndev) return fail(SP_EINVAL, "device %d out of range (have %d)", a->device, ndev);
    HIP_TRY(hipSetDevice(a->device));
    a->kernel_ms = 0.f;
    if (a->n_rows == 0 || (!a->out_rows && !a->out_cols_of_t)) return SP_OK;
    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    const int blocks = std::min(256 * 16, (a->n_rows + 255) / 256);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    DevPool pool;
    pool.device = a->device;
    const float *d_data = a->data;
    const int32_t *d_indptr = a->indptr;
    float *o_rows = a->out_rows, *o_cols = a->out_cols_of_t;
    hipStream_t stream = a->on_device ? (hipStream_t)a->stream : nullptr;
    if (!a->on_device) {
        TRY(pool.up(a->data, (size_t)a->nnz, &d_data));
        TRY(pool.up(a->indptr, (size_t)a->n_rows + 1, &d_indptr));
        if (a->out_rows) TRY(pool.alloc((size_t)a->n_rows, &o_rows));
        if (a->out_cols_of_t) TRY(pool.alloc((size_t)a->n_rows, &o_cols));
    }
    if (timed) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
    hipLaunchKernelGGL(sp_row_sqsums_kernel, dim3(blocks), dim3(256), 0, stream, a->n_rows, d_data, d_indptr, o_rows, o_cols);
    hipLaunchKernelGGL(sp_row_sqsums_long_kernel, dim3(std::min(a->n_rows, 2048)), dim3(256), 0, stream, a->n_rows, d_data, d_indptr, o_rows, o_cols);
    HIP_TRY(hipGetLastError());
    if (timed) {
        HIP_TRY(hipEventRecord(ev1, stream)); HIP_TRY(hipEventSynchronize(ev1)); HIP_TRY(hipEventElapsedTime(&a->kernel_ms, ev0, ev1));
        (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    }
    if (!a->on_device) {
        HIP_TRY(hipDeviceSynchronize());
        if (a->out_rows) HIP_TRY(hipMemcpy(a->out_rows, o_rows, (size_t)a->n_rows * 4, hipMemcpyDeviceToHost));
        if (a->out_cols_of_t) HIP_TRY(hipMemcpy(a->out_cols_of_t, o_cols, (size_t)a->n_rows * 4, hipMemcpyDeviceToHost));
    }
    return SP_OK;
}

int64_t sp_device_cache_trim(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    (void)hipDeviceSynchronize();
    return (int64_t)g_cache.trim(dev);
}

int sp_csr_col_sums_f32(sp_csr_colsums_args *a) {
    g_err[0] = 0;
    if (!a || a->struct_size != sizeof(sp_csr_colsums_args)) return fail(SP_EINVAL, "sp_csr_colsums_args size mismatch");
    if (a->n_cols < 0 || a->nnz < 0 || a->nnz > 0x7FFFFFFFLL) return fail(SP_EINVAL, "bad shape / nnz");
    if ((a->n_cols > 0 && !a->out) || (a->nnz > 0 && (!a->data || !a->indices))) return fail(SP_EINVAL, "NULL pointer");
    const int ndev = sp_device_count();
    if (ndev <= 0) return fail(SP_ENODEVICE, "no HIP device visible: similaripy_amd has no CPU fallback");
    if (a->device < 0 || a->device >= ndev) return fail(SP_EINVAL, "device %d out of range (have %d)", a->device, ndev);
    HIP_TRY(hipSetDevice(a->device));
    a->kernel_ms = 0.f;
    if (a->n_cols == 0) return SP_OK;
    DevPool pool;
    pool.device = a->device;
    const float *d_data = a->data;
    const int32_t *d_idx = a->indices;
    float *d_out = a->out;
    hipStream_t stream = a->on_device ? (hipStream_t)a->stream : nullptr;
    if (!a->on_device) {
        for (int64_t i = 0; i < a->nnz; ++i)
            if (a->indices[i] < 0 || a->indices[i] >= a->n_cols) return fail(SP_EINVAL, "indices[%lld]=%d out of range [0,%d)", (long long)i, a->indices[i], a->n_cols);
        TRY(pool.up(a->data, (size_t)a->nnz, &d_data));
        TRY(pool.up(a->indices, (size_t)a->nnz, &d_idx));
        TRY(pool.alloc((size_t)a->n_cols, &d_out));
    }
    double *acc = nullptr;
    TRY(pool.alloc((size_t)a->n_cols, &acc));
    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    CallGuard guard;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) { TRY(guard.event(&ev0)); TRY(guard.event(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
    HIP_TRY(hipMemsetAsync(acc, 0, (size_t)a->n_cols * 8, stream));
    if (a->nnz > 0) {
        if (a->square) hipLaunchKernelGGL((sp_col_sums_f64_kernel<true>), dim3(256 * 8), dim3(256), 0, stream, (long long)a->nnz, d_data, d_idx, acc);
        else hipLaunchKernelGGL((sp_col_sums_f64_kernel<false>), dim3(256 * 8), dim3(256), 0, stream, (long long)a->nnz, d_data, d_idx, acc);
    }
    hipLaunchKernelGGL(sp_f64_to_f32_kernel, dim3((a->n_cols + 255) / 256), dim3(256), 0, stream, a->n_cols, (const double *)acc, d_out);
    HIP_TRY(hipGetLastError());
    if (timed) { HIP_TRY(hipEventRecord(ev1, stream)); HIP_TRY(hipEventSynchronize(ev1)); HIP_TRY(hipEventElapsedTime(&a->kernel_ms, ev0, ev1)); }
    if (!a->on_device) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(a->out, d_out, (size_t)a->n_cols * 4, hipMemcpyDeviceToHost));
    }
    return SP_OK;
}

int sp_csr_normalize(sp_csr_normalize_args *a) {
    g_err[0] = 0;
    if (!a || a->struct_size != sizeof(sp_csr_normalize_args)) return fail(SP_EINVAL, "sp_csr_normalize_args size mismatch");
    if (a->n_rows < 0 || a->n_cols < 0 || a->nnz < 0 || a->nnz > 0x7FFFFFFFLL) return fail(SP_EINVAL, "bad shape / nnz");
    if (a->dtype != 0 && a->dtype != 1) return fail(SP_EINVAL, "dtype must be 0 (float32) or 1 (float64)");
    if (a->mode < SP_NORM_L1 || a->mode > SP_NORM_BM25PLUS) return fail(SP_EINVAL, "bad mode %d", a->mode);
    const bool weighted = a->mode == SP_NORM_TFIDF || a->mode == SP_NORM_BM25PLUS;
    if (weighted && (a->tf_mode < 0 || a->tf_mode > 4 || a->idf_mode < 0 || a->idf_mode > 4)) return fail(SP_EINVAL, "bad tf / idf mode");
    if (!a->indptr || (a->nnz > 0 && (!a->data || (weighted && !a->indices)))) return fail(SP_EINVAL, "NULL input pointer");
    const int ndev = sp_device_count();
    if (ndev <= 0) return fail(SP_ENODEVICE, "no HIP device visible: similaripy_amd has no CPU fallback");
    if (a->device < 0 || a->device >= ndev) return fail(SP_EINVAL, "device %d out of range (have %d)", a->device, ndev);
    HIP_TRY(hipSetDevice(a->device));
    a->kernel_ms = 0.f;
    if (a->n_rows == 0 || a->nnz == 0) return SP_OK;
    const size_t esz = a->dtype ? 8 : 4;
    DevPool pool;
    pool.device = a->device;
    void *d_data = a->data;
    const int32_t *d_indices = a->indices, *d_indptr = a->indptr;
    hipStream_t stream = a->on_device ? (hipStream_t)a->stream : nullptr;
    if (!a->on_device) {
        if (a->indptr[0] != 0 || (int64_t)a->indptr[a->n_rows] != a->nnz) return fail(SP_EINVAL, "indptr does not span [0, nnz]");
        for (int r = 0; r < a->n_rows; ++r) if (a->indptr[r + 1] < a->indptr[r]) return fail(SP_EINVAL, "indptr decreases at row %d", r);
        if (weighted) for (int64_t i = 0; i < a->nnz; ++i) if (a->indices[i] < 0 || a->indices[i] >= a->n_cols) return fail(SP_EINVAL, "indices[%lld] out of range", (long long)i);
        TRY(pool.raw((size_t)a->nnz * esz, &d_data));
        HIP_TRY(hipMemcpy(d_data, a->data, (size_t)a->nnz * esz, hipMemcpyHostToDevice));
        if (weighted) TRY(pool.up(a->indices, (size_t)a->nnz, &d_indices));
        TRY(pool.up(a->indptr, (size_t)a->n_rows + 1, &d_indptr));
    }
    // scratch of the weighted modes: document lengths, document frequencies, idf, average length
    void *doc_len = nullptr, *idf = nullptr, *avg = nullptr;
    int *df = nullptr;
    if (weighted) {
        TRY(pool.raw((size_t)a->n_rows * esz, &doc_len));
        TRY(pool.raw(std::max<size_t>(1, (size_t)a->n_cols) * esz, &idf));
        TRY(pool.raw(64, &avg));
        TRY(pool.alloc(std::max<size_t>(1, (size_t)a->n_cols), &df));
        HIP_TRY(hipMemsetAsync(df, 0, std::max<size_t>(1, (size_t)a->n_cols) * 4, stream));
    }
    const bool timed = (a->flags & SP_FLAG_TIME_KERNEL) != 0;
    CallGuard guard;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) { TRY(guard.event(&ev0)); TRY(guard.event(&ev1)); HIP_TRY(hipEventRecord(ev0, stream)); }
    const int wb = std::max(1, std::min(256 * 16, (a->n_rows + 3) / 4));
    auto run = [&](auto tag) {
        using T = decltype(tag);
        T *data = (T *)d_data;
        if (a->mode == SP_NORM_L1) hipLaunchKernelGGL((sp_row_normalize_kernel<T, RO_L1>), dim3(wb), dim3(256), 0, stream, a->n_rows, data, d_indptr, a->pow_alpha, (unsigned long long *)nullptr);
        else if (a->mode == SP_NORM_L2) hipLaunchKernelGGL((sp_row_normalize_kernel<T, RO_L2>), dim3(wb), dim3(256), 0, stream, a->n_rows, data, d_indptr, a->pow_alpha, (unsigned long long *)nullptr);
        else if (a->mode == SP_NORM_MAX) hipLaunchKernelGGL((sp_row_normalize_kernel<T, RO_MAX>), dim3(wb), dim3(256), 0, stream, a->n_rows, data, d_indptr, a->pow_alpha, (unsigned long long *)nullptr);
        else {
            // log_logbase = log(logbase) held in the data type (normalization.pyx:223, 296)
            const T llb = (T)log((double)(T)a->logbase);      // (the reference's float32 instantiation rounds logbase to float first: log((float)e) = 0.99999994)
            hipLaunchKernelGGL((sp_doc_stats_kernel<T>), dim3(wb), dim3(256), 0, stream, a->n_rows, (const T *)data, d_indices, d_indptr, (T *)doc_len, df);
            hipLaunchKernelGGL((sp_idf_kernel<T>), dim3((a->n_cols + 255) / 256), dim3(256), 0, stream, a->n_cols, (const int *)df, (T *)idf, a->n_rows, a->idf_mode, llb);
            hipLaunchKernelGGL((sp_avg_doc_len_kernel<T>), dim3(1), dim3(1024), 0, stream, a->n_rows, (const T *)doc_len, (T *)avg);
            if (a->mode == SP_NORM_TFIDF)
                hipLaunchKernelGGL((sp_tf_weight_kernel<T, RO_TFIDF>), dim3(wb), dim3(256), 0, stream, a->n_rows, data, d_indices, d_indptr, (const T *)doc_len,
                                   (const T *)idf, (const T *)avg, a->tf_mode, llb, (T)0, (T)0, (T)0);
            else
                hipLaunchKernelGGL((sp_tf_weight_kernel<T, RO_BM25PLUS>), dim3(wb), dim3(256), 0, stream, a->n_rows, data, d_indices, d_indptr, (const T *)doc_len,
                                   (const T *)idf, (const T *)avg, a->tf_mode, llb, (T)a->k1, (T)a->b, (T)a->delta);
        }
    };
    if (a->dtype) run(double()); else run(float());
    HIP_TRY(hipGetLastError());
    if (timed) { HIP_TRY(hipEventRecord(ev1, stream)); HIP_TRY(hipEventSynchronize(ev1)); HIP_TRY(hipEventElapsedTime(&a->kernel_ms, ev0, ev1)); }
    if (!a->on_device) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(a->data, d_data, (size_t)a->nnz * esz, hipMemcpyDeviceToHost));
    }
    return SP_OK;
}

int sp_knn_f32_i32(sp_knn_args *a) {
    g_err[0] = 0;
    int rc = validate(a);
    if (rc) return rc;
    const int ndev = sp_device_count();
    if (ndev <= 0) return fail(SP_ENODEVICE, "no HIP device visible: similaripy_amd has no CPU fallback");
    if (a->device < 0 || a->device >= ndev) return fail(SP_EINVAL, "device %d out of range (have %d)", a->device, ndev);
    if ((a->flags & SP_FLAG_REUSE_M2_PREP) && (!a->on_device || !a->workspace))      // (checked here, not in validate(): the workspace query runs without one)
        return fail(SP_EINVAL, "SP_FLAG_REUSE_M2_PREP needs device mode and the caller workspace of the call whose passes are reused");
    if (a->n_devices > 1) {
        if (a->on_device) return fail(SP_EINVAL, "n_devices > 1 is a host-mode option (device-resident operands live on ONE device)");
        if (a->n_targets == 0) return SP_OK;
        return run_host_multi(a);
    }
    if (a->n_devices == 1 && a->device_ids) {
        if (a->device_ids[0] < 0 || a->device_ids[0] >= ndev) return fail(SP_EINVAL, "device_ids[0] = %d out of range (have %d)", a->device_ids[0], ndev);
        const int32_t dev0 = a->device;
        a->device = a->device_ids[0];
        rc = a->on_device ? run_device(a) : run_host(a);
        a->device = dev0;
        return rc;
    }
    return a->on_device ? run_device(a) : run_host(a);
}

}  // extern "C"
