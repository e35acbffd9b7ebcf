// sp_rowops.hpp — the segmented row operations either side of the kernel, on the device:
//
//   row normalisers          similaripy/cython_code/normalization.pyx:97-334   l1 / l2 / max, tf-idf, BM25 / BM25+
//   p3alpha / rp3beta prep   similaripy/similarity.py:410-415, 477-483         rows / L1, then ^alpha
//   column sums              similaripy/cython_code/s_plus_utils.pyx:160-164   np.bincount (float64) of an explicit matrix2
//   CSR assembly             similaripy/cython_code/coo_to_csr.h:28-71, utils.pyx:141-173, s_plus.pyx:424
//                            slots -> counting sort by row -> eliminate_zeros
//
// All of it is HBM-bound streaming work over CSR arrays: one wave per row, lanes stride over the row (coalesced), two
// passes where a row statistic is needed first.  The reference's loops are sequential float sums (which its compiler
// vectorises under -ffast-math): sums here are wave reductions — the same values to within the rounding of a reordered
// float sum (a few ulp); everything elementwise follows the reference's mixed float/double arithmetic operation by
// operation (double literals promote, the result is rounded to the data type once).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

enum { RO_L1 = 0, RO_L2 = 1, RO_MAX = 2, RO_TFIDF = 3, RO_BM25PLUS = 4 };
enum { RO_TF_BINARY = 0, RO_TF_RAW, RO_TF_SQRT, RO_TF_FREQ, RO_TF_LOG };
enum { RO_IDF_UNARY = 0, RO_IDF_BASE, RO_IDF_SMOOTH, RO_IDF_PROB, RO_IDF_BM25 };

template <typename T>
__device__ __forceinline__ T ro_wave_sum(T v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T ro_wave_max(T v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const T o = __shfl_xor(v, d, 64); v = (o > v) ? o : v; }
    return v;
}
// a * b and a + b rounded separately (the reference's x86-64 baseline build has no fused multiply-add)
__device__ __forceinline__ float ro_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double ro_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float ro_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double ro_add(double a, double b) { return __dadd_rn(a, b); }

// number of stored entries equal to zero (+0.0 or -0.0): the reference drops them first (eliminate_zeros, s_plus.pyx:210-211)
__global__ __launch_bounds__(256) void sp_zero_count_kernel(long long nnz, const float *__restrict__ data, unsigned long long *__restrict__ counter) {
    unsigned n = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) n += (data[i] == 0.f) ? 1u : 0u;
    n = (unsigned)ro_wave_sum((int)n);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(counter, (unsigned long long)n);
}

// Structure check of an uploaded CSR (what scipy's check_format does on the host): st[0] |= 1 indptr[0] != 0, |= 2 indptr decreases
// (st[1] = first such row), |= 4 indptr[n_rows] != nnz; st[2] / st[3] = min / max column id.  st comes in as {0, INT_MAX, 0, -1}.
// Reads indptr[0..n_rows] and indices[0..nnz) only, whatever they hold.
__global__ __launch_bounds__(256) void sp_check_csr_kernel(int n_rows, long long nnz, const int *__restrict__ indptr, const int *__restrict__ indices, int *__restrict__ st) {
    const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x, nthr = (long long)gridDim.x * blockDim.x;
    int flags = 0, bad_row = 0x7FFFFFFF;
    if (t0 == 0 && n_rows > 0) flags |= (indptr[0] != 0 ? 1 : 0) | ((long long)indptr[n_rows] != nnz ? 4 : 0);
    for (long long r = t0; r < n_rows; r += nthr)
        if (indptr[r + 1] < indptr[r]) { flags |= 2; bad_row = min(bad_row, (int)r); }
    int lo = 0, hi = -1;
    for (long long i = t0; i < nnz; i += nthr) { const int c = indices[i]; lo = min(lo, c); hi = max(hi, c); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        flags |= __shfl_xor(flags, d, 64);
        bad_row = min(bad_row, __shfl_xor(bad_row, d, 64));
        lo = min(lo, __shfl_xor(lo, d, 64));
        hi = max(hi, __shfl_xor(hi, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (flags) atomicOr(&st[0], flags);
        if (flags & 2) atomicMin(&st[1], bad_row);
        if (lo < 0) atomicMin(&st[2], lo);
        if (hi >= 0) atomicMax(&st[3], hi);
    }
}

// number of rows whose column ids descend somewhere (one wave per row)
__global__ __launch_bounds__(256) void sp_rows_sorted_kernel(int n_rows, const int *__restrict__ indptr, const int *__restrict__ indices, unsigned int *__restrict__ bad_rows) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        int bad = 0;
        for (int i = b + 1 + lane; i < e; i += 64) bad |= (indices[i] < indices[i - 1]) ? 1 : 0;
        if (__any(bad) && lane == 0) atomicAdd(bad_rows, 1u);
    }
}

// ---- l1 / l2 / max (normalization.pyx:97-197), optionally followed by ^alpha (similarity.py:411, 413) ----
// One wave per row.  Rows whose norm is 0 (max: <= 0, or empty) are left alone.
// zero_made (optional): counts the stored non-zero entries that leave as 0.0 (underflow of the divide or the power) — the p3alpha / rp3beta
// preprocessing needs to know: the reference drops such entries afterwards (eliminate_zeros, s_plus.pyx:210-211)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void sp_row_normalize_kernel(int n_rows, T *__restrict__ data, const int *__restrict__ indptr, double pow_alpha,
                                                                unsigned long long *__restrict__ zero_made) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        if (e <= b) continue;
        T s;
        if (MODE == RO_MAX) {
            T m = data[b];
            for (int i = b + lane; i < e; i += 64) { const T x = data[i]; m = (x > m) ? x : m; }
            s = ro_wave_max(m);
            if (!(s > (T)0)) continue;
        } else {
            T acc = (T)0;
            for (int i = b + lane; i < e; i += 64) {
                const T x = data[i];
                acc += (MODE == RO_L1) ? (x < (T)0 ? -x : x) : ro_mul(x, x);
            }
            s = ro_wave_sum(acc);
            if (s == (T)0) continue;
            if (MODE == RO_L2) s = (T)sqrt((double)s);
        }
        unsigned made = 0;
        for (int i = b + lane; i < e; i += 64) {
            const T x0 = data[i];
            T x = x0 / s;
            if (pow_alpha != 1.0) x = (T)pow((double)x, pow_alpha);      // np.power in the data type: correctly rounded from double
            data[i] = x;
            made += (x == (T)0 && x0 != (T)0) ? 1u : 0u;
        }
        if (zero_made && __ballot(made != 0u)) {      // (rare: one atomic per row that has any)
            made = (unsigned)ro_wave_sum((int)made);
            if (lane == 0) atomicAdd(zero_made, (unsigned long long)made);
        }
    }
}

// ---- tf-idf and BM25 / BM25+ (normalization.pyx:200-334): rows = documents, columns = terms ----
// doc_len[i] = sum of the row; df[c] = number of rows with a positive entry in column c
template <typename T>
__global__ __launch_bounds__(256) void sp_doc_stats_kernel(int n_rows, const T *__restrict__ data, const int *__restrict__ indices,
                                                            const int *__restrict__ indptr, T *__restrict__ doc_len, int *__restrict__ df) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        T acc = (T)0;
        for (int i = b + lane; i < e; i += 64) {
            const T x = data[i];
            acc += x;
            if (x > (T)0) atomicAdd(&df[indices[i]], 1);
        }
        acc = ro_wave_sum(acc);
        if (lane == 0) doc_len[r] = acc;
    }
}

// avg[0] = (sum of doc_len) / n_docs in the data type (one workgroup)
template <typename T>
__global__ __launch_bounds__(1024) void sp_avg_doc_len_kernel(int n_rows, const T *__restrict__ doc_len, T *__restrict__ avg) {
    __shared__ double part[16];
    T acc = (T)0;
    for (int i = threadIdx.x; i < n_rows; i += 1024) acc += doc_len[i];
    acc = ro_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (double)acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T s = (T)0;
        for (int w = 0; w < 16; ++w) s += (T)part[w];
        avg[0] = n_rows > 0 ? s / (T)n_rows : (T)0;
    }
}

// idf[c] from df[c] (0 stays 0: normalization.pyx:243-245); log() is the C double function in the reference, its
// argument is formed in the data type
template <typename T>
__global__ __launch_bounds__(256) void sp_idf_kernel(int n_cols, const int *__restrict__ df, T *__restrict__ idf, int n_docs_i, int mode, T log_logbase) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const T f = (T)df[c], n_docs = (T)n_docs_i;
    T v = (T)0;
    if (df[c] != 0) {
        if (mode == RO_IDF_UNARY) v = (T)1;
        else if (mode == RO_IDF_BASE) v = (T)(log((double)(n_docs / f)) / (double)log_logbase);
        else if (mode == RO_IDF_SMOOTH) v = (T)(log((double)n_docs / (1.0 + (double)f)) / (double)log_logbase);      // (1 + x: int + float is a double in Cython)
        else if (mode == RO_IDF_PROB) v = (T)(log((double)((n_docs - f) / f)) / (double)log_logbase);
        else v = (T)(log(((double)(n_docs - f) + 0.5) / ((double)f + 0.5)) / (double)log_logbase);      // (+ 0.5: double literals)
    }
    idf[c] = v;
}

template <typename T>
__device__ __forceinline__ T ro_tf(T freq, T doc_len, int mode, T log_logbase) {
    if (mode == RO_TF_BINARY) return (freq != (T)0) ? (T)1 : (T)0;
    if (mode == RO_TF_RAW) return freq;
    if (mode == RO_TF_SQRT) return (T)sqrt((double)freq);
    if (mode == RO_TF_FREQ) return freq / doc_len;
    return (T)(log(1.0 + (double)freq) / (double)log_logbase);
}

// MODE RO_TFIDF:    data = tf * idf[col]
// MODE RO_BM25PLUS: data = idf[col] * (tf * (k1 + 1.0) / (tf + k1 * norm_len) + delta),  norm_len = (1.0 - b) + b * doc_len / avg
template <typename T, int MODE>
__global__ __launch_bounds__(256) void sp_tf_weight_kernel(int n_rows, T *__restrict__ data, const int *__restrict__ indices, const int *__restrict__ indptr,
                                                            const T *__restrict__ doc_len, const T *__restrict__ idf, const T *__restrict__ avg,
                                                            int tf_mode, T log_logbase, T k1, T bb, T delta) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const T avg_len = avg[0];
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        const T dl = doc_len[r];
        // (1.0 - b) is a double in the reference, b * doc_len / avg is formed in the data type, the sum is rounded to the data type
        const T norm_len = (T)((1.0 - (double)bb) + (double)(ro_mul(bb, dl) / avg_len));
        for (int i = b + lane; i < e; i += 64) {
            const T tf = ro_tf(data[i], dl, tf_mode, log_logbase);
            const T w = idf[indices[i]];
            if (MODE == RO_TFIDF) data[i] = ro_mul(tf, w);
            else {
                const T den = ro_add(tf, ro_mul(k1, norm_len));
                data[i] = (T)((double)w * (((double)tf * ((double)k1 + 1.0)) / (double)den + (double)delta));
            }
        }
    }
}

// out[r] = sum of row r (wave per row), float32
__global__ __launch_bounds__(256) void sp_row_sums_kernel(int n_rows, const float *__restrict__ data, const int *__restrict__ indptr, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        float acc = 0.f;
        for (int i = b + lane; i < e; i += 64) acc += data[i];
        acc = ro_wave_sum(acc);
        if (lane == 0) out[r] = acc;
    }
}

// ---- small elementwise helpers ----
// out[i] = in[i]^p in float32 (np.power(..., dtype=float32), s_plus_utils.pyx:257-276): correctly rounded from double
__global__ __launch_bounds__(256) void sp_pow_f32_kernel(int n, const float *__restrict__ in, float *__restrict__ out, double p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)pow((double)in[i], p);
}
// acc[indices[i]] += f(data[i]) in float64 (np.bincount's accumulator, s_plus_utils.pyx:160-164); SQUARE: f = x*x formed in float32
template <bool SQUARE>
__global__ __launch_bounds__(256) void sp_col_sums_f64_kernel(long long nnz, const float *__restrict__ data, const int *__restrict__ indices, double *__restrict__ acc) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const float x = data[i];
        atomicAdd(&acc[indices[i]], (double)(SQUARE ? __fmul_rn(x, x) : x));
    }
}
__global__ __launch_bounds__(256) void sp_f64_to_f32_kernel(int n, const double *__restrict__ in, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---- CSR assembly of the kernel's slots (coo_to_csr.h:28-71 + eliminate_zeros), fast form for strictly increasing targets ----
// row_nnz[targets[i] + 1] = number of NON-ZERO values among the first counts[i] entries of slot i   (row_nnz pre-zeroed, n_rows + 1 long;
// an exclusive... inclusive scan over it then gives indptr directly)
__global__ __launch_bounds__(256) void sp_slot_nnz_kernel(int n_targets, int k, const int *__restrict__ targets, const int *__restrict__ counts,
                                                           const float *__restrict__ values, int *__restrict__ row_nnz_shifted) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave0; s < n_targets; s += n_waves) {
        const int n = counts[s];
        const float *v = values + s * (long long)k;
        int c = 0;
        for (int j = lane; j < n; j += 64) c += (v[j] != 0.f) ? 1 : 0;
        c = ro_wave_sum(c);
        if (lane == 0) row_nnz_shifted[targets[s] + 1] = c;
    }
}

// Chunked CSR assembly (strictly increasing targets: slot order IS row order): non-zero entries per slot of a chunk of slots, and their
// compaction to the front of the chunk's own output range at the slots' scanned offsets — what lets a chunk's rows travel to the host
// while the next chunk's rows are still being computed (run_host).
__global__ __launch_bounds__(256) void sp_chunk_slot_nnz_kernel(int n_slots, int k, const int *__restrict__ counts, const float *__restrict__ values, int *__restrict__ slot_nnz) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave0; s < n_slots; s += n_waves) {
        const int n = counts[s];
        const float *v = values + s * (long long)k;
        int c = 0;
        for (int j = lane; j < n; j += 64) c += (v[j] != 0.f) ? 1 : 0;
        c = ro_wave_sum(c);
        if (lane == 0) slot_nnz[s] = c;
    }
}
__global__ __launch_bounds__(256) void sp_chunk_compact_kernel(int n_slots, int k, const int *__restrict__ counts, const int *__restrict__ cols, const float *__restrict__ values,
                                                                const int *__restrict__ slot_off, int *__restrict__ out_indices, float *__restrict__ out_data) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave0; s < n_slots; s += n_waves) {
        const int n = counts[s];
        const long long src = s * (long long)k;
        long long dst = slot_off[s];
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const float v = (j < n) ? values[src + j] : 0.f;
            const int c = (j < n) ? cols[src + j] : 0;
            const unsigned long long m = __ballot(v != 0.f);
            if (v != 0.f) {
                const long long q = dst + __popcll(m & ((1ull << lane) - 1ull));
                out_indices[q] = c;
                out_data[q] = v;
            }
            dst += __popcll(m);
        }
    }
}

// non-zero entries of slot i, in slot order, to [indptr[targets[i]], ...)
__global__ __launch_bounds__(256) void sp_csr_compact_kernel(int n_targets, int k, const int *__restrict__ targets, const int *__restrict__ counts,
                                                              const int *__restrict__ cols, const float *__restrict__ values, const int *__restrict__ indptr,
                                                              int *__restrict__ out_indices, float *__restrict__ out_data) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave0; s < n_targets; s += n_waves) {
        const int n = counts[s];
        const long long src = s * (long long)k;
        long long dst = indptr[targets[s]];
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const float v = (j < n) ? values[src + j] : 0.f;
            const int c = (j < n) ? cols[src + j] : 0;
            const unsigned long long m = __ballot(v != 0.f);
            if (v != 0.f) {
                const long long q = dst + __popcll(m & ((1ull << lane) - 1ull));
                out_indices[q] = c;
                out_data[q] = v;
            }
            dst += __popcll(m);
        }
    }
}

// ---- ARRAY column selectors on an explicit m2 (_filter_matrix_columns, s_plus_utils.pyx:424-490): entries whose column has
// keep[c] == 0 are dropped, the order inside a row stays.  One wave per row: kept entries per row (shifted by one: an inclusive
// scan gives the new row pointers), then the compaction. ----
__global__ __launch_bounds__(256) void sp_keep_count_kernel(int n_rows, const int *__restrict__ indptr, const int *__restrict__ indices,
                                                             const unsigned char *__restrict__ keep, int *__restrict__ row_nnz_shifted) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        int c = 0;
        for (int i = b + lane; i < e; i += 64) c += keep[indices[i]] ? 1 : 0;
        c = ro_wave_sum(c);
        if (lane == 0) row_nnz_shifted[r + 1] = c;
    }
}

__global__ __launch_bounds__(256) void sp_keep_compact_kernel(int n_rows, const int *__restrict__ indptr, const int *__restrict__ indices,
                                                               const float *__restrict__ data, const unsigned char *__restrict__ keep,
                                                               const int *__restrict__ new_indptr, int *__restrict__ out_indices, float *__restrict__ out_data) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        const int b = indptr[r], e = indptr[r + 1];
        int dst = new_indptr[r];
        for (int i0 = b; i0 < e; i0 += 64) {
            const int i = i0 + lane;
            const int c = (i < e) ? indices[i] : 0;
            const bool kept = (i < e) && keep[c] != 0;
            const unsigned long long m = __ballot(kept);
            if (kept) {
                const int q = dst + __popcll(m & ((1ull << lane) - 1ull));
                out_indices[q] = c;
                out_data[q] = data[i];
            }
            dst += __popcll(m);
        }
    }
}

// ---- the same for ANY order of the targets (unsorted, repeated: `target_rows=[7, 2, 7]`): coo_to_csr.h:28-71 is a STABLE counting
// sort of the slots by row, so the entries of a row that is asked for twice appear slot after slot.  Per slot: its non-zero
// count, the row's total (atomic) and the number of slots of its row; per row with more than one slot: its slots in slot order
// (a scatter with an atomic cursor, then an insertion sort of the handful of slot ids) and each slot's offset inside the row. ----
__global__ __launch_bounds__(256) void sp_slot_nnz_any_kernel(int n_targets, int k, const int *__restrict__ targets, const int *__restrict__ counts,
                                                               const float *__restrict__ values, int *__restrict__ slot_nnz,
                                                               int *__restrict__ row_nnz_shifted, int *__restrict__ row_slots_shifted) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave0; s < n_targets; s += n_waves) {
        const int n = counts[s];
        const float *v = values + s * (long long)k;
        int c = 0;
        for (int j = lane; j < n; j += 64) c += (v[j] != 0.f) ? 1 : 0;
        c = ro_wave_sum(c);
        if (lane == 0) {
            slot_nnz[s] = c;
            atomicAdd(&row_nnz_shifted[targets[s] + 1], c);
            atomicAdd(&row_slots_shifted[targets[s] + 1], 1);
        }
    }
}

// bucket[bstart[t] + (arrival order)] = slot id, for every slot of row t
__global__ __launch_bounds__(256) void sp_slot_scatter_kernel(int n_targets, const int *__restrict__ targets, const int *__restrict__ bstart,
                                                               int *__restrict__ cursor, int *__restrict__ bucket) {
    for (long long s = blockIdx.x * (long long)blockDim.x + threadIdx.x; s < n_targets; s += (long long)gridDim.x * blockDim.x) {
        const int t = targets[s];
        bucket[bstart[t] + atomicAdd(&cursor[t], 1)] = (int)s;
    }
}

// slot_off[slot] = non-zero entries of the EARLIER slots of the same row (one thread per row; rows asked for once: 0)
__global__ __launch_bounds__(256) void sp_slot_offsets_kernel(int n_rows, const int *__restrict__ bstart, int *__restrict__ bucket,
                                                               const int *__restrict__ slot_nnz, int *__restrict__ slot_off) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n_rows; t += (long long)gridDim.x * blockDim.x) {
        const int b0 = bstart[t], b1 = bstart[t + 1];
        if (b1 - b0 == 1) { slot_off[bucket[b0]] = 0; continue; }
        for (int i = b0 + 1; i < b1; ++i) {          // insertion sort by slot id: the scatter's arrival order is arbitrary
            const int x = bucket[i];
            int j = i - 1;
            while (j >= b0 && bucket[j] > x) { bucket[j + 1] = bucket[j]; --j; }
            bucket[j + 1] = x;
        }
        int run = 0;
        for (int i = b0; i < b1; ++i) { slot_off[bucket[i]] = run; run += slot_nnz[bucket[i]]; }
    }
}

// non-zero entries of slot i, in slot order, to [indptr[targets[i]] + slot_off[i], ...)
__global__ __launch_bounds__(256) void sp_csr_compact_any_kernel(int n_targets, int k, const int *__restrict__ targets, const int *__restrict__ counts,
                                                                  const int *__restrict__ cols, const float *__restrict__ values, const int *__restrict__ indptr,
                                                                  const int *__restrict__ slot_off, int *__restrict__ out_indices, float *__restrict__ out_data) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave0; s < n_targets; s += n_waves) {
        const int n = counts[s];
        const long long src = s * (long long)k;
        long long dst = (long long)indptr[targets[s]] + slot_off[s];
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const float v = (j < n) ? values[src + j] : 0.f;
            const int c = (j < n) ? cols[src + j] : 0;
            const unsigned long long m = __ballot(v != 0.f);
            if (v != 0.f) {
                const long long q = dst + __popcll(m & ((1ull << lane) - 1ull));
                out_indices[q] = c;
                out_data[q] = v;
            }
            dst += __popcll(m);
        }
    }
}

}  // namespace
