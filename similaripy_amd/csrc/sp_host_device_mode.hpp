// sp_host_device_mode.hpp — host side of the C ABI, part 3 (textually included by sp_knn.hip inside its anonymous namespace): device-mode calls that
// build m2 (or m1) on the device first, the sampled route for target_cols = <matrix>, chunked sub-launches (run_device).
// (no include guard on purpose: it is one file's text, cut out for reading — not a header of declarations)
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
