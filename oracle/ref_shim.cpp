// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// extern "C" entry point around the REFERENCE kernel itself.  This file holds
// no algorithm: it includes similaripy/cython_code/s_plus.h from where it lies
// under /root/reference (passed with -I by oracle/Makefile; never copied into
// this repository) and instantiates
//   s_plus::compute_similarities_parallel<int,float>      (s_plus.h:265-453)
// exactly as the reference's Cython seam does (s_plus.pyx:359-384), with
// progress == nullptr (s_plus.h:340-342 skips the bar then).
//
// Output: oracle/_ref/libsplus_ref.so (git-ignored; travels to the GPU box with
// the gpurun snapshot).  Used to pin oracle/splus_port.c and, in bench.py, as
// the CPU baseline of kind "reference".
#include "similaripy/cython_code/s_plus.h"

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" void splus_ref_compute(
    int n_targets, const int* targets,
    const float* m1_data, const int* m1_indices, const int* m1_indptr,
    const float* m2_data, const int* m2_indices, const int* m2_indptr,
    const float* Xtv, const float* Ytv, const float* Xcos, const float* Ycos,
    const float* Xdep, const float* Ydep,
    float a1, float l1, float l2, float l3, float t1, float t2,
    float stab, float bayes, float threshold,
    int k, int n_output_cols,
    int filter_mode, const int* f_indptr, const int* f_indices,
    int target_mode, const int* t_indptr, const int* t_indices,
    int* rows, int* cols, float* values,
    int num_threads, int block_size)
{
#ifdef _OPENMP
    if (num_threads <= 0) num_threads = omp_get_max_threads();
#else
    num_threads = 1;
#endif
    s_plus::compute_similarities_parallel<int, float>(
        n_targets, targets,
        m1_data, m1_indices, m1_indptr,
        m2_data, m2_indices, m2_indptr,
        Xtv, Ytv, Xcos, Ycos, Xdep, Ydep,
        a1, l1, l2, l3, t1, t2, stab, bayes, threshold,
        k, n_output_cols,
        filter_mode, const_cast<int*>(f_indptr), const_cast<int*>(f_indices),
        target_mode, const_cast<int*>(t_indptr), const_cast<int*>(t_indices),
        rows, cols, values,
        nullptr, num_threads, block_size);
}

extern "C" int splus_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
