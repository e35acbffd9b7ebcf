// sp_host_multi.hpp — host side of the C ABI, part 5 (textually included by sp_knn.hip inside its anonymous namespace): the ONE partition cost model
// (target_costs / partition_by_cost) and the host-mode call over several devices (run_host_multi).
// (no include guard on purpose: it is one file's text, cut out for reading — not a header of declarations)
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
