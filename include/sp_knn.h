/*
 * sp_knn.h — C ABI of libsimilaripy_hip.so: the MI355X (gfx950) replacement for
 * the one native kernel of bogliosimone/similaripy.
 *
 * What it replaces in the reference
 * ---------------------------------
 *   s_plus::compute_similarities_parallel<int,float>   similaripy/cython_code/s_plus.h:265-453
 *   declared to Cython at                               similaripy/cython_code/s_plus.pyx:51-91
 *   called (GIL released) at                            similaripy/cython_code/s_plus.pyx:359-384
 *
 * sp_knn_f32_i32() takes the same arguments with the same meaning (CSR m1, CSR
 * m2, the six normalisation vectors, the nine scalars, k, the two column
 * selectors, the three flat output arrays of n_targets*k entries).  Because a
 * device copy needs sizes the reference never passes, the struct additionally
 * carries n_rows_m1 / n_rows_m2 / nnz counts.  Output convention is the
 * reference's (s_plus.h:444-450): slot i owns [k*i, k*i+k); the first n_i <= k
 * entries hold (row = targets[i], col, value) in unspecified order and the tail
 * is zero.  Unlike the reference the callee zero-fills the tail itself, so the
 * caller need not pre-zero.
 *
 * No torch / scipy / C++ types cross this boundary: plain pointers and sizes.
 * All functions return 0 on success or a negative SP_E* code; sp_last_error()
 * gives the thread's last message.  (The reference kernel is `void`, has no
 * error path and is UB on bad input — s_plus.pyx:51; here bad input is
 * reported instead.)
 */
#ifndef SP_KNN_H_
#define SP_KNN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* column selector modes — s_plus.h:24-28 (enum SelectionMode) */
#define SP_SEL_NONE   0
#define SP_SEL_ARRAY  1   /* pre-applied to m2 by the host; the kernel treats it as NONE (s_plus.h:160-162) */
#define SP_SEL_MATRIX 2   /* per-row sorted column lists, indexed by ABSOLUTE m1 row id (s_plus.h:165-169) */

/* error codes */
#define SP_OK            0
#define SP_EINVAL       -1   /* bad argument / struct size mismatch */
#define SP_ENODEVICE    -2   /* no HIP device: the product path never falls back to CPU */
#define SP_EHIP         -3   /* a HIP runtime call failed (see sp_last_error) */
#define SP_ENOMEM       -4
#define SP_EWORKSPACE   -5   /* caller workspace too small */
#define SP_EZEROS       -6   /* SP_FLAG_CHECK_ZEROS: the matrices hold explicit zeros (see explicit_zeros); nothing was computed */
#define SP_EUNDERFLOW   -8   /* SP_FLAG_P3_PREP (host mode): stored entries underflowed to 0.0 in the L1 divide or the power (explicit_zeros holds the
                                count).  The reference drops them before its kernel runs (similarity.py:410-415, then eliminate_zeros,
                                s_plus.pyx:210-211); the outputs of this call still hold them as zero-valued candidates and must be discarded:
                                the caller preprocesses on the host and calls again without SP_FLAG_P3_PREP */
#define SP_EUNSORTED    -7   /* host mode: a row of m2 (SP_FLAG_M1_IS_M2_T / SP_FLAG_CHECK_SORTED) does not have ascending column ids; nothing was computed */
#define SP_EUNSORTED_SELECTOR -9  /* host mode: a row of a MATRIX selector (always checked) does not have ascending column ids; nothing was computed.
                                (A code of its own since round 6: the caller's remedy differs — sort the selector, not m2 — and must not hang on the message text) */

/* flags */
#define SP_FLAG_TIME_KERNEL   1u  /* bracket device work with hipEvents on `stream`, sync, fill kernel_ms */
#define SP_FLAG_NO_ROWS_OUT   2u  /* `rows` may be NULL and is not written (it is targets[i] repeated over slot i: a CSR consumer does not need it) */
#define SP_FLAG_STATIC_SCHED  4u  /* round-robin rows over workgroups instead of the atomic row queue */
#define SP_FLAG_NO_SPARSE_PATH 8u /* never use the bitmap + collision-set path for sparse rows (A/B testing) */
#define SP_FLAG_NO_FOLD       16u /* never divide the column term into the m2 stream (A/B testing) */
#define SP_FLAG_NO_ROW_ORDER  32u /* queue rows in target order instead of descending work (A/B testing) */
#define SP_FLAG_M2_IS_M1_T    128u /* m2 = m1^T (the `matrix2=None` call, s_plus.pyx:169-170): built on the device from m1 by the callee;
                                     the m2_* pointers and nnz_m2 are ignored (may be NULL / 0), n_rows_m2 = columns of m1,
                                     n_output_cols must equal n_rows_m1.  See sp_prep.h. */
#define SP_FLAG_PHASE_TIMERS  64u /* with SP_FLAG_TIME_KERNEL: also run the in-kernel phase timers (s_memtime, ~1-2 % slower) */
/* host mode (on_device = 0) only: the stages of s_plus.pyx that the reference runs on the host around its kernel */
#define SP_FLAG_CHECK_ZEROS   256u /* count the stored entries of m1 (and of an explicit m2) that are 0 on the device, after the upload
                                     (the reference calls eliminate_zeros() first, s_plus.pyx:210-211); if there are any, nothing is
                                     computed, explicit_zeros holds the count and SP_EZEROS is returned: the caller drops them and calls again */
#define SP_FLAG_CSR_OUT       512u /* assemble the CSR result on the device (build_csr_matrix utils.pyx:141-173 -> coo_to_csr.h:28-71 ->
                                     eliminate_zeros s_plus.pyx:424): any order of `targets`, repeats included (a row asked for twice holds its
                                     slots one after the other, as the reference's stable counting sort leaves them); csr_indptr receives the
                                     n_rows_m1 + 1 row pointers, the first csr_nnz entries of `cols` / `values` the column ids and values of
                                     the non-zero entries in row order (slot order inside a row); rows / out_counts are not written.
                                     With a MATRIX target selector AND strictly ascending `targets` the result cannot hold more than target_col_nnz
                                     entries: `cols` / `values` then need only min(n_targets * k, target_col_nnz) entries when the call runs on ONE
                                     device (nothing beyond csr_nnz is written or touched).  Any other order of `targets` — a repeated target emits
                                     its row once per repeat — needs the full n_targets * k. */
#define SP_FLAG_P3_PREP      1024u /* with SP_FLAG_M2_IS_M1_T or SP_FLAG_M1_IS_M2_T: the preprocessing of p3alpha / rp3beta (similarity.py:410-415, 477-483;
                                     normalization.pyx:131-161) on the device: the rows of m1 and the rows of m2 = m1^T are divided by
                                     their L1 norms, then every entry is raised to p3_alpha.  The caller's matrix is not modified. */
#define SP_FLAG_DEPOP_ROWSUM 2048u /* with SP_FLAG_P3_PREP and l3 != 0: Ydepop[c] = (sum of the RAW row c of m1)^depop_p2, i.e. the column
                                     popularity of the raw m2 = m1^T (similarity.py:479; np.power in float32, s_plus_utils.pyx:257-276),
                                     built on the device; the Ydepop pointer is ignored */
/* ABI 3 */
#define SP_FLAG_M1_IS_M2_T   4096u /* m1 = m2^T, built on the device from m2: the call whose matrix1 arrives as CSC (`sim.cosine(URM.T)`, the
                                     documented item-item usage; the reference converts it with matrix1.tocsr() on the host,
                                     s_plus.pyx:205-206, while matrix2 = matrix1.T already is the CSR the caller holds).  The m1_*
                                     pointers and nnz_m1 are ignored, n_rows_m1 must equal n_output_cols, the rows of m2 must have
                                     ascending column ids (host mode checks: SP_EUNSORTED).  Excludes SP_FLAG_M2_IS_M1_T. */
#define SP_FLAG_NORMS_ON_DEVICE 8192u /* with SP_FLAG_M2_IS_M1_T or SP_FLAG_M1_IS_M2_T: the Xtversky / Ytversky (l1 != 0) and Xcosine /
                                     Ycosine (l2 != 0) pointers are ignored; the vectors are built on the device from the rows of m1
                                     (_build_squared_norms s_plus_utils.pyx:169-201 as sp_csr_row_sqsums_f32 does, sp_prep.h;
                                     _build_cosine_normalization :204-228 with norm_c1 / norm_c2 / norm_add).
                                     Host mode also with an explicit m2: the row sums of m1^2 and the column sums of m2^2 are formed
                                     from the uploaded copies (sp_csr_row_sqsums_f32 / sp_csr_col_sums_f32 on device pointers) — the
                                     host layer uploaded m2 a second time for them */

/* ABI 5 */
#define SP_FLAG_REUSE_M2_PREP 16384u /* device mode with a caller workspace: the workspace still holds the per-call passes over m2 / Y* of an
                                     earlier call (column term folded into the m2 values or the packed column terms and their minima, the
                                     dense-window boundaries inside every m2 row, the sign flag) and they are not redone.  The caller vouches
                                     that m2, the Y* vectors, every scalar parameter, k and the tuning fields are those of that call and that
                                     nothing else wrote the workspace in between; only `targets` / n_targets and the outputs may differ (they
                                     must not need a larger workspace than that call had).  This is how one step over a slice of target rows
                                     is cut into sub-launches whose results travel while the next one computes (distributed.py). */

#define SP_FLAG_BINARY      32768u /* host mode only: `binary=True` (s_plus.pyx:214-217, m.data = ones AFTER eliminate_zeros): the stored values of
                                     m1 (and of an explicit m2) are replaced by 1.0 in the uploaded copies, after the SP_FLAG_CHECK_ZEROS count
                                     has seen the caller's values.  The caller's arrays are not modified (and no array of ones is built or
                                     uploaded: 256 MB at 64 M entries).  Device-mode callers own their buffers and fill them themselves. */
#define SP_FLAG_CHECK_SORTED 65536u /* host mode, explicit m2: the rows of the uploaded m2 are checked for ascending column ids on the device (the
                                     requirement above); a descent anywhere: SP_EUNSORTED, nothing computed — the caller sorts and calls again
                                     (what the host layer did with a pass over m2's indices before every call) */

#define SP_FLAG_PROGRESS    131072u /* host mode: one line on stderr whenever a chunk of result rows has reached the host ("rows done a / n") — the coarse
                                     counterpart of ProgressBar::update (progress_bar.h:199-208, driven from s_plus.h:340-342); `verbose=True` sets it */

typedef struct sp_knn_args {
    uint32_t struct_size;      /* = sizeof(sp_knn_args); checked */
    uint32_t flags;            /* SP_FLAG_* */
    int32_t  on_device;        /* 0: every pointer below is host memory (the drop-in call: H2D, compute, D2H)
                                  1: every pointer is device memory on `device`, resident before the call; the launches are queued on
                                     `stream` and the call returns without waiting (every call: the one exception of rounds 3-5 — a
                                     depopularisation-only epilogue read a 4-byte flag back before going on — is gone since round 6, the
                                     fold itself writes the zero the reference's zero denominator gives, s_plus.h:144-150): the call can be
                                     stream-captured */
    int32_t  device;           /* HIP device ordinal */

    /* problem shape (the reference infers these from NumPy arrays it never passes down) */
    int32_t  n_targets;        /* s_plus.h: n_targets */
    int32_t  n_rows_m1;        /* rows of m1 (length of m1_indptr - 1, of X* vectors, of selector indptr - 1) */
    int32_t  n_rows_m2;        /* rows of m2 == columns of m1 */
    int32_t  n_output_cols;    /* columns of m2 (length of Y* vectors) */
    int64_t  nnz_m1;
    int64_t  nnz_m2;

    const int32_t *targets;    /* [n_targets] absolute m1 row ids, any order (s_plus.pyx:191-196) */
    const float   *m1_data;    const int32_t *m1_indices;   const int32_t *m1_indptr;
    const float   *m2_data;    const int32_t *m2_indices;   const int32_t *m2_indptr;
                               /* m2 rows must have ascending column ids whenever n_output_cols exceeds the
                                  accumulator tile (multi-pass), the same requirement the reference's blocked
                                  path has (s_plus.h:385-394).  The host layer guarantees it. */

    /* normalisation vectors; each pair is read only if the matching weight is non-zero
       (s_plus.h:134-139) and may be NULL / dangling otherwise (s_plus.pyx:248-256) */
    const float *Xtversky, *Ytversky;   /* l1 != 0 : sum x^2 per m1 row / per m2 column */
    const float *Xcosine,  *Ycosine;    /* l2 != 0 : (sum x^2 + additive_shrink)^c */
    const float *Xdepop,   *Ydepop;     /* l3 != 0 : w^p */

    float a1, l1, l2, l3, t1, t2;
    float stabilized_shrink, bayesian_shrink, threshold;

    int32_t k;                 /* 1 <= k */
    int32_t filter_mode;       /* SP_SEL_* */
    const int32_t *filter_m_indptr;      /* [n_rows_m1+1] when MATRIX */
    const int32_t *filter_m_indices;     /* sorted within a row */
    int64_t filter_nnz;
    int32_t target_col_mode;
    int32_t _pad0;
    const int32_t *target_col_m_indptr;
    const int32_t *target_col_m_indices;
    int64_t target_col_nnz;

    /* outputs, n_targets*k each (64-bit offsets inside: n_targets*k may exceed 2^31) */
    int32_t *rows;
    int32_t *cols;
    float   *values;
    int32_t *out_counts;       /* optional [n_targets]: n_i per slot (the reference returns none) */

    /* device-mode plumbing */
    void    *stream;           /* hipStream_t; NULL = the null stream */
    void    *workspace;        /* device scratch of >= sp_knn_workspace_bytes(); NULL = allocate internally */
    int64_t  workspace_bytes;

    /* tuning, 0 = auto */
    int32_t table_slots;       /* LDS accumulator slots per workgroup (power of two) */
    int32_t threads_per_wg;    /* 256 / 512 / 1024 */
    int32_t num_wgs;           /* persistent workgroups */
    int32_t load_pct;          /* hash fill target in percent (default 50) */

    /* results */
    float   kernel_ms;         /* OUT when SP_FLAG_TIME_KERNEL */
    int32_t passes_total;      /* OUT (debug, only with SP_FLAG_TIME_KERNEL): accumulate+drain passes summed over rows */
    int64_t phase_cycles[12];  /* OUT with SP_FLAG_TIME_KERNEL | SP_FLAG_PHASE_TIMERS: shader cycles summed over workgroups (lane 0), both row kernels:
                                  [0] row setup  [1] generic: segment scan | sparse: item list, bitmap clear + rank prefix
                                  [2] generic: accumulate | sparse: member products into the collision set
                                  [3] judge (column terms, epilogue, top-k buffer)  [4] selections  [5] write-out
                                  [6] sparse sweep 1  [7] sparse sweep 2  [8] bit 0: the sparse rows ran on the wave-per-row kernel (light rows), else on the
                                  workgroup-per-row kernel; bit 1: that kernel's bounded variant ran (general epilogue, column-term code in the
                                  m2 ids); then event counts:
                                  [9] rows finished by the sparse kernel  [10] rows it handed to the generic kernel
                                  [11] generic column windows */
    int32_t num_wgs_used;      /* OUT with SP_FLAG_TIME_KERNEL */
    int32_t _pad1;
    int64_t reserved[4];       /* [0] IN: kernel ablation bits, profiling / tests only (0 in production) — THE table (one place):
                                    ROUTE switches (results stay correct; tests use them to keep the other route covered):
                                       1024  force the generic kernel's 64-bit-offset variant for every row (as nnz(m2) >= 2^30 does;
                                             tests/test_hip_parity.py::test_generic_kernel_64bit_offset_variant)
                                       2048  no per-call work-item prepass (rows are set up in the kernel; test_sparse_kernel_packed_trips*)
                                       4096  heavy generic rows are NOT queued in pieces
                                       8192  cut EVERY generic row into pieces as finely as allowed (test_generic_kernel_heavy_rows_in_pieces)
                                      16384  no wave-per-row kernel
                                      32768  no bounded variant: general epilogues on the general variant (test_bounded_variant_* A/B)
                                      65536  no sampled (SDDMM) route for target_cols = <matrix> (test_target_matrix_sampled_route*,
                                             test_repeated_targets_with_a_target_matrix_and_csr_out)
                                     131072  generic kernel: no selection-free cutoff in front of a dense window's drain
                                     262144  generic kernel: the drain judges in the sweep (round-4 form) instead of sorting live slots first
                                    2097152  generic kernel: the selection-free cutoff pass runs in front of EVERY dense window's drain (it stops once the row has a cutoff)
                                    4194304  generic kernel: a light row's next window / a heavy row's next batch of m1 entries is NOT requested ahead (scripts/c4_heavy_probe.py A/B;
                                             test_generic_kernel_window_chain_ablations)
                                    1048576  sparse kernel, two-per-CU shape: no second launch in the larger layout (rows over the first's limit go to the generic kernel)
                                     524288  sparse kernel: no two-per-CU (DUO) shape (scripts/c2_phases.py A/B; test_duo_shape_against_the_oracle_and_the_classic_shape)
                                    PROFILING ablations (results are WRONG; the sweep-body ones compiled in only with -DSP_ABLATION=1):
                                          1  generic accumulate without LDS inserts      4  no column-term gathers
                                          8 / 16  sweep 1 / sweep 2 load but do not process      32 / 64  members / survivors dropped
                                        128  sparse kernel: workgroups start staggered; emit_candidates: no epilogue
                                        256  sparse kernel: a loads-only pass in front of sweep 1; generic kernel: no simple judge
                                        512  generic kernel: no register selection; wave kernel: every trip reads the same 4 KB
                                      32768  (wave kernel) sweep 2's id loads hit the cache
                                  [1], [2] OUT with SP_FLAG_TIME_KERNEL: duration of the sparse / generic row kernel of this
                                  call in microseconds (hipEvents on `stream` around each launch)
                                  [3] OUT with SP_FLAG_TIME_KERNEL | SP_FLAG_M2_IS_M1_T: duration of the transpose, microseconds
                                  (kernel_ms includes it) */

    /* ABI 2: host-side stages of s_plus.pyx on the device */
    float    p3_alpha;         /* SP_FLAG_P3_PREP */
    float    depop_p2;         /* SP_FLAG_DEPOP_ROWSUM */
    int32_t *csr_indptr;       /* SP_FLAG_CSR_OUT: OUT [n_rows_m1 + 1], caller-allocated */
    int64_t  csr_nnz;          /* SP_FLAG_CSR_OUT: OUT */
    int64_t  explicit_zeros;   /* SP_FLAG_CHECK_ZEROS: OUT */

    /* ABI 3 */
    float    norm_c1, norm_c2; /* SP_FLAG_NORMS_ON_DEVICE: Xcosine = (sum x^2 + norm_add)^norm_c1, Ycosine = (sum y^2 + norm_add)^norm_c2 */
    float    norm_add;         /* additive_shrink */
    int32_t  _pad2;

    /* ABI 4 */
    const uint8_t *col_keep;   /* optional: [n_output_cols] bytes; output columns c with col_keep[c] == 0 are dropped from m2 before the
                                  row kernels run — the ARRAY form of filter_cols / target_cols (compute_target_columns +
                                  _filter_matrix_columns, s_plus_utils.pyx:364-490), which the reference applies to matrix2 on the host.
                                  With SP_FLAG_M2_IS_M1_T (host or device pointers): the rows of m1 with a 0 are skipped while m2 = m1^T
                                  is built.  With an explicit m2: host mode only (the uploaded copy is compacted; a device-resident m2
                                  belongs to the caller).  With SP_FLAG_P3_PREP (and SP_FLAG_M2_IS_M1_T): the columns are dropped from the NORMALISED m2, as the
                                  reference does it (similarity.py:410-415 before s_plus_utils.pyx:424-490).  Not with SP_FLAG_M1_IS_M2_T.
                                  NULL: every column stays. */

    /* ABI 5: several devices behind ONE host-mode call */
    int32_t  n_devices;        /* host mode only.  0 or 1: the call runs on `device`.  N > 1: `targets` is cut into N contiguous slices of
                                  equal cost (MACs(t) + a fixed toll per row: what the row kernels schedule on) and slice r runs on
                                  device_ids[r] — one host thread per device, every device gets its own copy of the operands over its own
                                  PCIe link (m2 / Y* replicated, as in the one-process-per-GPU layout), no exchange during compute, and
                                  every device writes its slots straight into the caller's output arrays.  The row loop being sharded is
                                  the reference's own parallel loop (s_plus.h:313, 337).  With SP_FLAG_CSR_OUT the targets must ascend
                                  strictly (each device assembles the CSR rows of its slice; the pieces are joined on the host). */
    int32_t  _pad3;
    const int32_t *device_ids; /* [n_devices] distinct HIP ordinals; NULL = 0 .. n_devices-1 */
} sp_knn_args;

/* The hot path.  Replaces compute_similarities_parallel<int,float> (s_plus.h:265). */
int sp_knn_f32_i32(sp_knn_args *args);

/* Device scratch the call needs for these args (device mode with caller workspace). */
int64_t sp_knn_workspace_bytes(const sp_knn_args *args);

/* The partition cost model of the multi-GPU routes, in ONE place ("contiguous ranges balanced by sum MACs(t)"; the row loop
   being cut is s_plus.h:313, 337).  Host-mode args (on_device = 0); only the CSR STRUCTURE of m1 / m2, `targets`, the sizes and the flags
   SP_FLAG_M2_IS_M1_T / SP_FLAG_M1_IS_M2_T are read — no device is needed.
   sp_knn_target_costs: cost[i] of target slot i in MAC equivalents = MACs + a per-row toll (30 k for a row of the sparse kernels, 3 per
   output column for a row of the generic kernel) + a price per m1 entry of a heavy row the launch cuts into pieces.
   sp_knn_partition: bounds[0 .. n_parts] of contiguous slices of equal cumulative cost — what n_devices > 1 cuts `targets` by, and what
   similaripy_amd/distributed.py (one process per GPU) calls for its ranks. */
int sp_knn_target_costs(const sp_knn_args *args, double *cost /* [n_targets] */);
int sp_knn_partition(const sp_knn_args *args, int n_parts, int64_t *bounds /* [n_parts + 1] */);

/* Number of usable HIP devices (0 when there is none).  Counterpart of
   get_num_threads() -> omp_get_max_threads()  (similaripy/cython_code/utils.pyx:18-25). */
int sp_device_count(void);

/* Human-readable backend description ("gfx950 ... 256 CUs ..."); returns bytes written or <0. */
int sp_backend_info(int device, char *buf, int buflen);

/* Last error message of the calling thread ("" if none). */
const char *sp_last_error(void);

/* The library keeps the device buffers of host-mode calls (operands, outputs, workspace) in a size-bucketed cache instead of
   returning them to the driver after every call (hipMalloc / hipFree of GB-sized buffers cost milliseconds each).  This
   releases the cache of the calling thread's current device; returns the bytes released. */
int64_t sp_device_cache_trim(void);

/* ABI version of this header. */
#define SP_KNN_ABI_VERSION 5
int sp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SP_KNN_H_ */
