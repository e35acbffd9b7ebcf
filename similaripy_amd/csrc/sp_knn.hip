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
#include "sp_host_config.hpp"

#include "sp_host_launch.hpp"

}  // namespace
#include "sp_transpose.hpp"
namespace {

#include "sp_host_device_mode.hpp"

#include "sp_host_mode.hpp"

#include "sp_host_multi.hpp"

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
