// sp_host_launch.hpp — host side of the C ABI, part 2 (textually included by sp_knn.hip inside its anonymous namespace): the launches of the row
// kernels, the signature of the per-call passes over m2, and run_device_impl — one call with every pointer on the device.
// (no include guard on purpose: it is one file's text, cut out for reading — not a header of declarations)
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
