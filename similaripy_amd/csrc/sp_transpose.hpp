// sp_transpose.hpp — CSR transpose on the device: m2 = m1^T for the `matrix2=None` call.
//
// Replaces, for float32 / int32 CSR, what the reference does on the host with scipy before its kernel runs:
//   matrix2 = matrix1.T                      similaripy/cython_code/s_plus.pyx:169-170   (a CSC view)
//   matrix2 = matrix2.tocsr()                similaripy/cython_code/s_plus.pyx:205-206   (scipy csc_tocsr: counting sort)
// Result: the CSR of the transpose with ascending column ids inside every row — bit-identical to scipy's
// `m.T.tocsr()` of a canonical CSR (for an input row holding a column twice, scipy keeps input order between the
// two; here they are ordered by value bits: both are "the same multiset", the kernel adds them up either way).
//
// Four launches, all HBM-bound integer work:
//   count     one atomic per non-zero on its column's counter
//   scan      exclusive prefix over the counters -> row pointers of the transpose (+ a second copy: the cursors)
//   scatter   one wave per input row: position = atomic on the column's cursor; {row id, value bits} as one
//             64-bit record (arrival order within an output row is whatever the atomics made it)
//   sort      every output row is sorted by row id — bitonic in LDS (<= 1024 / <= 16384 records), in global memory
//             beyond — and split into the index and value arrays.  This makes the result deterministic.
#pragma once

namespace {

typedef unsigned long long tr_u64;

__global__ __launch_bounds__(256) void sp_tr_count_kernel(long long nnz, const int *__restrict__ indices, int *__restrict__ cnt) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x)
        atomicAdd(&cnt[indices[i]], 1);
}

// the same with a mask over the INPUT rows (keep_rows[r] == 0: row r contributes nothing): one wave per input row
__global__ __launch_bounds__(256) void sp_tr_count_rows_kernel(int n_rows, const int *__restrict__ indices, const int *__restrict__ indptr,
                                                                const unsigned char *__restrict__ keep_rows, int *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        if (!keep_rows[r]) continue;
        const int b = indptr[r], e = indptr[r + 1];
        for (int i = b + lane; i < e; i += 64) atomicAdd(&cnt[indices[i]], 1);
    }
}

// one wave per input row
__global__ __launch_bounds__(256) void sp_tr_scatter_kernel(int n_rows, const float *__restrict__ data, const int *__restrict__ indices,
                                                             const int *__restrict__ indptr, const unsigned char *__restrict__ keep_rows,
                                                             int *__restrict__ cursor, tr_u64 *__restrict__ rec) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = wave0; r < n_rows; r += n_waves) {
        if (keep_rows != nullptr && !keep_rows[r]) continue;
        const int b = indptr[r], e = indptr[r + 1];
        for (int i = b + lane; i < e; i += 64) {
            const int pos = atomicAdd(&cursor[indices[i]], 1);
            rec[pos] = ((tr_u64)(unsigned)r << 32) | (tr_u64)__float_as_uint(data[i]);
        }
    }
}

// In-place ascending sort of buf[0..len) by all `nt` threads of the workgroup: the bitonic network in its all-ascending
// form (first step of a merge pairs i with its mirror in the block, the others i with i+j; every comparator leaves the
// smaller record at the lower index).  The padding up to the next power of two is virtual: a comparator whose upper
// end lies beyond `len` would compare with +inf and never swap, so it is skipped.
template <typename Ptr>
__device__ __forceinline__ void tr_bitonic(Ptr buf, int len, int tid, int nt) {
    int lp = 1;
    while ((1ll << lp) < len) ++lp;
    const long long half = 1ll << (lp - 1);
    for (int lk = 1; lk <= lp; ++lk) {           // merge blocks of k = 2^lk records
        const long long k = 1ll << lk, hk = k >> 1;
        for (long long t = tid; t < half; t += nt) {
            const long long blk = t >> (lk - 1), off = t & (hk - 1);
            const long long lo = (blk << lk) + off, hi = (blk << lk) + (k - 1 - off);
            if (hi < len) {
                const tr_u64 a = buf[lo], c = buf[hi];
                if (a > c) { buf[lo] = c; buf[hi] = a; }
            }
        }
        __syncthreads();
        for (long long j = k >> 2; j > 0; j >>= 1) {
            for (long long t = tid; t < half; t += nt) {
                const long long lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                if (hi < len) {
                    const tr_u64 a = buf[lo], c = buf[hi];
                    if (a > c) { buf[lo] = c; buf[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// rows of the transpose with min_len < length <= MAX_LEN: sort in LDS (MAX_LEN records of dynamic shared memory)
template <int MAX_LEN>
__global__ __launch_bounds__(256) void sp_tr_sort_lds_kernel(int n_out_rows, int min_len, const int *__restrict__ out_indptr, const tr_u64 *__restrict__ rec,
                                                              int *__restrict__ out_indices, float *__restrict__ out_data) {
    extern __shared__ tr_u64 tr_buf[];
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < n_out_rows; r += gridDim.x) {
        const int b = out_indptr[r], len = out_indptr[r + 1] - b;
        if (len <= min_len || len > MAX_LEN) continue;       // (uniform) another launch's class
        for (int i = tid; i < len; i += 256) tr_buf[i] = rec[b + i];
        __syncthreads();
        if (len > 1) tr_bitonic(tr_buf, len, tid, 256);
        for (int i = tid; i < len; i += 256) {
            const tr_u64 v = tr_buf[i];
            out_indices[b + i] = (int)(unsigned)(v >> 32);
            out_data[b + i] = __uint_as_float((unsigned)v);
        }
        __syncthreads();
    }
}

// rows longer than the LDS classes (a popular item of a ratings matrix: 10^5 entries): the same network with LDS tiles.
// Tiles of TR_TILE records are sorted in LDS; a merge level then runs its long-distance steps (mirror step, j >= TR_TILE) on the
// records in global memory and its short ones (j < TR_TILE, all inside aligned tiles) tile by tile in LDS again: 15 passes over
// the row at 2^18 records instead of 171.  One workgroup per row: its own stores are visible to it after the barrier.
constexpr int TR_TILE = 8192;
__global__ __launch_bounds__(1024) void sp_tr_sort_global_kernel(int n_out_rows, int min_len, const int *__restrict__ out_indptr, tr_u64 *rec,
                                                                  int *__restrict__ out_indices, float *__restrict__ out_data) {
    extern __shared__ tr_u64 tr_buf[];
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < n_out_rows; r += gridDim.x) {
        const int b = out_indptr[r], len = out_indptr[r + 1] - b;
        if (len <= min_len) continue;       // (uniform)
        tr_u64 *buf = rec + b;
        int lp = 1;
        while ((1ll << lp) < len) ++lp;
        const long long half = 1ll << (lp - 1);
        int lt = 1;
        while ((1 << lt) < TR_TILE) ++lt;
        // levels 1 .. lt: every tile sorted on its own
        for (int t0 = 0; t0 < len; t0 += TR_TILE) {
            const int n = min(TR_TILE, len - t0);
            for (int i = tid; i < n; i += 1024) tr_buf[i] = buf[t0 + i];
            __syncthreads();
            if (n > 1) tr_bitonic(tr_buf, n, tid, 1024);
            for (int i = tid; i < n; i += 1024) buf[t0 + i] = tr_buf[i];
            __syncthreads();
        }
        for (int lk = lt + 1; lk <= lp; ++lk) {      // merge blocks of k = 2^lk records
            const long long k = 1ll << lk, hk = k >> 1;
            for (long long t = tid; t < half; t += 1024) {
                const long long blk = t >> (lk - 1), off = t & (hk - 1);
                const long long lo = (blk << lk) + off, hi = (blk << lk) + (k - 1 - off);
                if (hi < len) {
                    const tr_u64 a = buf[lo], c = buf[hi];
                    if (a > c) { buf[lo] = c; buf[hi] = a; }
                }
            }
            __syncthreads();
            for (long long j = k >> 2; j >= TR_TILE; j >>= 1) {
                for (long long t = tid; t < half; t += 1024) {
                    const long long lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    if (hi < len) {
                        const tr_u64 a = buf[lo], c = buf[hi];
                        if (a > c) { buf[lo] = c; buf[hi] = a; }
                    }
                }
                __syncthreads();
            }
            // the steps j = TR_TILE/2 .. 1 pair records of one aligned tile
            for (int t0 = 0; t0 < len; t0 += TR_TILE) {
                const int n = min(TR_TILE, len - t0);
                for (int i = tid; i < n; i += 1024) tr_buf[i] = buf[t0 + i];
                __syncthreads();
                for (int j = TR_TILE >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < TR_TILE / 2; t += 1024) {
                        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                        if (hi < n) {
                            const tr_u64 a = tr_buf[lo], c = tr_buf[hi];
                            if (a > c) { tr_buf[lo] = c; tr_buf[hi] = a; }
                        }
                    }
                    __syncthreads();
                }
                for (int i = tid; i < n; i += 1024) buf[t0 + i] = tr_buf[i];
                __syncthreads();
            }
        }
        for (int i = tid; i < len; i += 1024) {
            const tr_u64 v = buf[i];
            out_indices[b + i] = (int)(unsigned)(v >> 32);
            out_data[b + i] = __uint_as_float((unsigned)v);
        }
        __syncthreads();
    }
}

// ---- squared row norms -----------------------------------------------------------------------------------------
// Replaces, for m2 = m1^T, _build_squared_norms (similaripy/cython_code/s_plus_utils.pyx:169-201) with csr_sum
// (:128-166), bit for bit:
//   out_rows[r]  = np.add.reduceat(data^2, indptr[:-1])[r]       float32, NumPy's pairwise summation: the first element
//                  plus pairwise(rest) — blocks of <= 128 elements with eight running sums, halves split at a
//                  multiple of eight above that (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum);
//                  an empty row gives 0 (s_plus_utils.pyx:155-158)
//   out_cols[r]  = float32(np.bincount(.., weights=data^2))      float64 running sum in storage order: the column sums
//                  of m1^T are the row sums of m1
// data^2 is rounded to float32 first (np.square(.., dtype=float32)); no fused multiply-add anywhere.
// x*x rounded to float32, opaque to the compiler: HIP's default -ffp-contract=fast would otherwise fuse the square into
// the addition that follows (even through __fmul_rn / __fadd_rn), which rounds once instead of twice
__device__ __forceinline__ float sq_rn(float x) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ float np_pairwise_sq(const float *a, long long n) {
    if (n < 8) {
        float res = -0.0f;
        for (long long i = 0; i < n; ++i) res = __fadd_rn(res, sq_rn(a[i]));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = sq_rn(a[j]);
        long long i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], sq_rn(a[i + j]));
        }
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, sq_rn(a[i]));
        return res;
    }
    long long n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_sq(a, n2), np_pairwise_sq(a + n2, n - n2));
}

constexpr int SQ_LONG = 4096;      // rows beyond this many entries get a workgroup of their own

// one thread per row (rows are short next to the row count in most shapes the reference is used on)
__global__ __launch_bounds__(256) void sp_row_sqsums_kernel(int n_rows, const float *__restrict__ data, const int *__restrict__ indptr,
                                                             float *__restrict__ out_rows, float *__restrict__ out_cols) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n_rows; r += (long long)gridDim.x * blockDim.x) {
        const int b = indptr[r], e = indptr[r + 1];
        if (e - b > SQ_LONG) continue;       // sp_row_sqsums_long_kernel's
        if (out_rows) {
            float v = 0.f;
            if (e > b) {
                v = sq_rn(data[b]);
                if (e - b > 1) v = __fadd_rn(v, np_pairwise_sq(data + b + 1, (long long)(e - b - 1)));
            }
            out_rows[r] = v;
        }
        if (out_cols) {
            double acc = 0.0;
            for (int i = b; i < e; ++i) acc = __dadd_rn(acc, (double)sq_rn(data[i]));
            out_cols[r] = (float)acc;
        }
    }
}

// Long rows (a popular item of a ratings matrix: 10^5..10^6 entries): one workgroup per row.
//   out_rows: NumPy's pairwise recursion is a binary tree; its top eight levels are laid out in LDS (node i: children 2i,
//   2i+1; a node of <= 128 elements is a leaf), every thread sums one level-8 subtree (or a shallower leaf) with the
//   same recursion, and the partial sums are added back up the same tree: the same float32 additions in the same
//   order as NumPy — bit-identical.
//   out_cols_of_t: np.bincount's float64 running sum is inherently sequential; here every thread sums a contiguous
//   chunk in float64 and the 256 partial sums are added in order.  The float64 roundings differ from the strictly
//   sequential sum (relative 1e-16); the float32 result differs only if that crosses a float32 rounding boundary (about
//   once in 1e8 rows).
__global__ __launch_bounds__(256) void sp_row_sqsums_long_kernel(int n_rows, const float *__restrict__ data, const int *__restrict__ indptr,
                                                                  float *__restrict__ out_rows, float *__restrict__ out_cols) {
    __shared__ long long n_off[512];
    __shared__ long long n_len[512];      // 0 = no such node
    __shared__ float n_val[512];
    __shared__ double part[256];
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int b = indptr[r], e = indptr[r + 1];
        if (e - b <= SQ_LONG) continue;       // (uniform)
        const float *a = data + b;
        const long long n = e - b;
        if (out_rows) {
            for (int i = tid; i < 512; i += 256) { n_len[i] = 0; n_off[i] = 0; }
            __syncthreads();
            if (tid == 0) { n_off[1] = 1; n_len[1] = n - 1; }       // (the first element is added last, as reduceat does)
            __syncthreads();
            for (int lvl = 0; lvl < 8; ++lvl) {
                const int i = (1 << lvl) + tid;
                if (tid < (1 << lvl) && n_len[i] > 128) {
                    long long n2 = n_len[i] / 2;
                    n2 -= n2 % 8;
                    n_off[2 * i] = n_off[i]; n_len[2 * i] = n2;
                    n_off[2 * i + 1] = n_off[i] + n2; n_len[2 * i + 1] = n_len[i] - n2;
                }
                __syncthreads();
            }
            // leaves of the eight-level tree: nodes without children (level 8 nodes, or <= 128 elements above)
            for (int i = 1 + tid; i < 512; i += 256) {
                const bool is_leaf = n_len[i] > 0 && (i >= 256 || n_len[2 * i] == 0);
                if (is_leaf) n_val[i] = np_pairwise_sq(a + n_off[i], n_len[i]);
            }
            __syncthreads();
            for (int lvl = 7; lvl >= 0; --lvl) {
                const int i = (1 << lvl) + tid;
                if (tid < (1 << lvl) && n_len[i] > 0 && n_len[2 * i] > 0) n_val[i] = __fadd_rn(n_val[2 * i], n_val[2 * i + 1]);
                __syncthreads();
            }
            if (tid == 0) out_rows[r] = __fadd_rn(sq_rn(a[0]), n_val[1]);
        }
        if (out_cols) {
            const long long per = (n + 255) / 256;
            const long long lo = min(n, tid * per), hi = min(n, lo + per);
            double acc = 0.0;
            for (long long i = lo; i < hi; ++i) acc = __dadd_rn(acc, (double)sq_rn(a[i]));
            part[tid] = acc;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int i = 0; i < 256; ++i) tot = __dadd_rn(tot, part[i]);
                out_cols[r] = (float)tot;
            }
        }
        __syncthreads();
    }
}

constexpr int TR_SHORT = 1024, TR_MEDIUM = 16384;

// bytes of scratch the transpose needs: counters/cursors (n_cols+1 ints each), the 64-bit records, the scan's chunk sums
size_t transpose_ws_bytes(long long nnz, int n_cols) {
    return (((size_t)n_cols + 1) * 4 * 2 + 255 & ~(size_t)255) + (((size_t)nnz * 8 + 255) & ~(size_t)255) + SCAN_SCRATCH_BYTES;
}

// all pointers on the device; asynchronous on `stream`.  keep_rows (optional, [n_rows] bytes): input rows with a 0 are left out —
// i.e. the COLUMNS of the result they would have become are empty: the ARRAY form of filter_cols / target_cols applied to
// m2 = m1^T (_filter_matrix_columns, s_plus_utils.pyx:424-490) costs nothing when m2 is built here.  The output arrays then
// hold fewer than nnz entries; their tails are zeroed (flat passes over "nnz(m2)" entries read them).
int transpose_device(int n_rows, int n_cols, long long nnz, const float *data, const int *indices, const int *indptr,
                     float *out_data, int *out_indices, int *out_indptr, void *ws, size_t ws_bytes, hipStream_t stream,
                     const unsigned char *keep_rows = nullptr) {
    if (ws_bytes < transpose_ws_bytes(nnz, n_cols)) return fail(SP_EWORKSPACE, "transpose workspace too small");
    int *cnt = (int *)ws;
    int *cursor = cnt + ((size_t)n_cols + 1);
    tr_u64 *rec = (tr_u64 *)((unsigned char *)ws + (((size_t)n_cols + 1) * 4 * 2 + 255 & ~(size_t)255));
    HIP_TRY(hipMemsetAsync(cnt, 0, ((size_t)n_cols + 1) * 4, stream));
    if (nnz > 0 && keep_rows != nullptr) {
        HIP_TRY(hipMemsetAsync(out_data, 0, (size_t)nnz * 4, stream));
        HIP_TRY(hipMemsetAsync(out_indices, 0, (size_t)nnz * 4, stream));
        const int blocks = (int)std::min<long long>(256 * 32, ((long long)n_rows + 3) / 4);
        hipLaunchKernelGGL(sp_tr_count_rows_kernel, dim3(blocks), dim3(256), 0, stream, n_rows, indices, indptr, keep_rows, cnt);
    } else if (nnz > 0) {
        const int blocks = (int)std::min<long long>(256 * 16, (nnz + 255) / 256);
        hipLaunchKernelGGL(sp_tr_count_kernel, dim3(blocks), dim3(256), 0, stream, nnz, indices, cnt);
    }
    // out_indptr[i] = entries before output row i (out_indptr[n_cols] = nnz); cursor = a second copy the scatter advances
    scan_i32<false>(n_cols, cnt, out_indptr, cursor, nullptr, (unsigned char *)rec + (((size_t)nnz * 8 + 255) & ~(size_t)255), stream);
    if (nnz > 0 && n_rows > 0) {
        const int blocks = (int)std::min<long long>(256 * 32, ((long long)n_rows + 3) / 4);
        hipLaunchKernelGGL(sp_tr_scatter_kernel, dim3(blocks), dim3(256), 0, stream, n_rows, data, indices, indptr, keep_rows, cursor, rec);
        const int sb = std::min(n_cols, 256 * 32);
        hipLaunchKernelGGL(sp_tr_sort_lds_kernel<TR_SHORT>, dim3(sb), dim3(256), TR_SHORT * 8, stream, n_cols, 0, out_indptr, rec, out_indices, out_data);
        // (per device, and cheap: set on every call like the row kernels' launchers do)
        HIP_TRY(hipFuncSetAttribute((const void *)sp_tr_sort_lds_kernel<TR_MEDIUM>, hipFuncAttributeMaxDynamicSharedMemorySize, TR_MEDIUM * 8));
        hipLaunchKernelGGL(sp_tr_sort_lds_kernel<TR_MEDIUM>, dim3(std::min(n_cols, 256 * 4)), dim3(256), TR_MEDIUM * 8, stream, n_cols, TR_SHORT, out_indptr, rec, out_indices, out_data);
        HIP_TRY(hipFuncSetAttribute((const void *)sp_tr_sort_global_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TR_TILE * 8));
        hipLaunchKernelGGL(sp_tr_sort_global_kernel, dim3(std::min(n_cols, 256 * 2)), dim3(1024), TR_TILE * 8, stream, n_cols, TR_MEDIUM, out_indptr, rec, out_indices, out_data);
    }
    HIP_TRY(hipGetLastError());
    return SP_OK;
}

}  // namespace
