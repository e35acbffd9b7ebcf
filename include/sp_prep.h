/*
 * sp_prep.h — C ABI of the device-side preprocessing that libsimilaripy_hip.so offers next to the kernel
 * (include/sp_knn.h): SURVEY.md §8(f) row 1, "on-device preprocessing".
 *
 * What it replaces in the reference
 * ---------------------------------
 *   matrix2 = matrix1.T            similaripy/cython_code/s_plus.pyx:169-170
 *   matrix2 = matrix2.tocsr()      similaripy/cython_code/s_plus.pyx:205-206   (scipy csc_tocsr on the host)
 * i.e. the transpose the reference builds with scipy whenever `matrix2` is not given (every `sim.cosine(m)`,
 * `sim.jaccard(m)`, ... call).  sp_csr_transpose_f32_i32() builds it on the GPU; sp_knn_f32_i32() with
 * SP_FLAG_M2_IS_M1_T (sp_knn.h) builds it internally and never materialises m2 on the host.
 *   _build_squared_norms           similaripy/cython_code/s_plus_utils.pyx:169-201   (np.add.reduceat / np.bincount)
 * sp_csr_row_sqsums_f32() gives both norm vectors of that call from the rows of m1.
 *
 * Result convention: CSR of the transpose, int32 indices ascending inside each row, float32 data — what
 * scipy returns for a canonical (sorted, duplicate-free) CSR input, bit for bit.
 * Same conventions as sp_knn.h: plain pointers and sizes, 0 or a negative SP_E* code, sp_last_error().
 */
#ifndef SP_PREP_H_
#define SP_PREP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sp_csr_transpose_args {
    uint32_t struct_size;      /* = sizeof(sp_csr_transpose_args); checked */
    uint32_t flags;            /* SP_FLAG_TIME_KERNEL (sp_knn.h) or 0 */
    int32_t  on_device;        /* 0: every pointer below is host memory (H2D, transpose, D2H); 1: device memory on `device` */
    int32_t  device;

    int32_t  n_rows;           /* input shape: n_rows x n_cols */
    int32_t  n_cols;
    int64_t  nnz;              /* < 2^31 */
    const float   *data;       /* [nnz] */
    const int32_t *indices;    /* [nnz], each in [0, n_cols) */
    const int32_t *indptr;     /* [n_rows + 1] */

    float   *out_data;         /* [nnz] */
    int32_t *out_indices;      /* [nnz]: input row ids, ascending inside each output row */
    int32_t *out_indptr;       /* [n_cols + 1] */

    void    *stream;           /* device mode: hipStream_t (NULL = the null stream); the call is asynchronous */
    void    *workspace;        /* device mode: scratch of >= sp_csr_transpose_workspace_bytes(); NULL = allocate internally (synchronous) */
    int64_t  workspace_bytes;
    float    kernel_ms;        /* OUT with SP_FLAG_TIME_KERNEL: all launches of the transpose, hipEvents on `stream` */
    int32_t  _pad0;
} sp_csr_transpose_args;

/* matrix.T.tocsr() for float32 / int32 CSR (s_plus.pyx:169-170, 205-206). */
int sp_csr_transpose_f32_i32(sp_csr_transpose_args *args);

/* Device scratch the call needs (8 bytes per non-zero + 8 bytes per column). */
int64_t sp_csr_transpose_workspace_bytes(const sp_csr_transpose_args *args);

typedef struct sp_csr_sqsums_args {
    uint32_t struct_size;      /* = sizeof(sp_csr_sqsums_args); checked */
    uint32_t flags;            /* SP_FLAG_TIME_KERNEL or 0 */
    int32_t  on_device;        /* 0: host pointers (H2D, kernel, D2H); 1: device pointers, asynchronous on `stream` */
    int32_t  device;
    int32_t  n_rows;
    int32_t  _pad0;
    int64_t  nnz;
    const float   *data;       /* [nnz] */
    const int32_t *indptr;     /* [n_rows + 1] */
    float   *out_rows;         /* [n_rows] or NULL: sum of data^2 per row as np.add.reduceat gives it (float32, NumPy's pairwise order) */
    float   *out_cols_of_t;    /* [n_rows] or NULL: the same sums as np.bincount gives them for the columns of the TRANSPOSE
                                  (float64 running sum in storage order, rounded to float32; rows beyond 4096 entries are
                                  summed in 256 float64 chunks: the float32 result can differ by one ulp about once in 1e8 rows) */
    void    *stream;
    float    kernel_ms;        /* OUT with SP_FLAG_TIME_KERNEL */
    int32_t  _pad1;
} sp_csr_sqsums_args;

/* The squared norms of the `matrix2=None` call: _build_squared_norms (similaripy/cython_code/s_plus_utils.pyx:169-201)
   = csr_sum(m1^2, axis=1) and csr_sum((m1^T)^2, axis=0) (:128-166), both from the rows of m1, bit-identical to NumPy. */
int sp_csr_row_sqsums_f32(sp_csr_sqsums_args *args);

/* Column sums of a CSR as np.bincount forms them (float64 accumulator, rounded to float32 at the end):
 *   csr_sum(axis=0)                       similaripy/cython_code/s_plus_utils.pyx:160-164
 * used by _build_squared_norms (:169-201) for the columns of an explicit matrix2 (square = 1: sums of data^2, the square
 * formed in float32 as np.square does) and by _build_depop_normalization (:231-278) for weight 'sum' (square = 0).
 * The float64 additions happen in whatever order the atomics arrive: the float32 result is np.bincount's except for
 * the rare sum whose float64 rounding error straddles a float32 rounding boundary (one ulp). */
typedef struct sp_csr_colsums_args {
    uint32_t struct_size;      /* = sizeof(sp_csr_colsums_args); checked */
    uint32_t flags;            /* SP_FLAG_TIME_KERNEL or 0 */
    int32_t  on_device;        /* 0: host pointers; 1: device pointers, asynchronous on `stream` (scratch from the library's cache) */
    int32_t  device;
    int32_t  n_cols;
    int32_t  square;           /* 1: sum of data^2, 0: sum of data */
    int64_t  nnz;
    const float   *data;       /* [nnz] */
    const int32_t *indices;    /* [nnz], each in [0, n_cols) */
    float   *out;              /* [n_cols] */
    void    *stream;
    float    kernel_ms;        /* OUT with SP_FLAG_TIME_KERNEL */
    int32_t  _pad0;
} sp_csr_colsums_args;

int sp_csr_col_sums_f32(sp_csr_colsums_args *args);

/* ---- row normalisers ------------------------------------------------------------------------------
 * In-place weighting of the rows of a CSR, the device counterpart of
 *   inplace_normalize_csr_l1 / _l2 / _max      similaripy/cython_code/normalization.pyx:97-197
 *   inplace_normalize_csr_tfidf                similaripy/cython_code/normalization.pyx:200-262
 *   inplace_normalize_csr_bm25plus             similaripy/cython_code/normalization.pyx:265-334   (BM25 = delta 0)
 * float32 or float64 data (the reference keeps either, normalization.py:23-40), int32 indices.  Sums are formed in the
 * data type like the reference's, in a different order (wave reductions): results agree to a few ulp. */
#define SP_NORM_L1       0
#define SP_NORM_L2       1
#define SP_NORM_MAX      2
#define SP_NORM_TFIDF    3
#define SP_NORM_BM25PLUS 4
/* tf / idf modes in the order of the reference's enums (normalization.pyx:12-24) */
#define SP_TF_BINARY 0
#define SP_TF_RAW    1
#define SP_TF_SQRT   2
#define SP_TF_FREQ   3
#define SP_TF_LOG    4
#define SP_IDF_UNARY  0
#define SP_IDF_BASE   1
#define SP_IDF_SMOOTH 2
#define SP_IDF_PROB   3
#define SP_IDF_BM25   4

typedef struct sp_csr_normalize_args {
    uint32_t struct_size;      /* = sizeof(sp_csr_normalize_args); checked */
    uint32_t flags;            /* SP_FLAG_TIME_KERNEL or 0 */
    int32_t  on_device;        /* 0: host pointers (H2D, kernels, D2H of data); 1: device pointers, asynchronous on `stream` */
    int32_t  device;
    int32_t  n_rows;           /* documents */
    int32_t  n_cols;           /* terms */
    int64_t  nnz;
    int32_t  dtype;            /* 0 = float32, 1 = float64 */
    int32_t  mode;             /* SP_NORM_* */
    void          *data;       /* [nnz] IN / OUT */
    const int32_t *indices;    /* [nnz]; read by TFIDF / BM25PLUS only (may be NULL otherwise) */
    const int32_t *indptr;     /* [n_rows + 1] */
    int32_t  tf_mode;          /* SP_TF_*  (TFIDF / BM25PLUS) */
    int32_t  idf_mode;         /* SP_IDF_* (TFIDF / BM25PLUS) */
    double   k1, b, delta;     /* BM25PLUS (BM25: delta = 0) */
    double   logbase;          /* TFIDF / BM25PLUS */
    double   pow_alpha;        /* L1 / L2 / MAX: entries are raised to this power afterwards when != 1 (similarity.py:411, 413) */
    void    *stream;           /* device mode */
    float    kernel_ms;        /* OUT with SP_FLAG_TIME_KERNEL */
    int32_t  _pad0;
} sp_csr_normalize_args;

int sp_csr_normalize(sp_csr_normalize_args *args);

#ifdef __cplusplus
}
#endif
#endif /* SP_PREP_H_ */
