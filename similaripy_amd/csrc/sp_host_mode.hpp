// sp_host_mode.hpp — host side of the C ABI, part 4 (textually included by sp_knn.hip inside its anonymous namespace): host-mode calls — the device
// buffer cache, staging, uploads, chunks whose results leave while the kernel runs (run_host).
// (no include guard on purpose: it is one file's text, cut out for reading — not a header of declarations)
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
