// Probe (gfx950): sustained rate of RANDOM 4-byte LDS operations over a 128 KiB region, 16 waves per CU, 8 independent
// operations per wave and wait: returning atomic OR (sweep 1), plain read (sweep 2), non-returning OR, 8-byte write.
// build: hipcc --offload-arch=gfx950 -O3 scripts/lds_random_probe.hip -o /tmp/ldsr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
template <int MODE>
__global__ __launch_bounds__(1024) void k(u64 *out, int iters) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += 1024) lds[i] = 0;
    __syncthreads();
    unsigned acc = 0, r = tid * 2654435761u + 7u;
    const u64 t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        unsigned a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { r = r * 1664525u + 1013904223u; a[q] = (r >> 17) & 32767u; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) acc += atomicOr(&lds[a[q]], 1u << (r & 31));
            if (MODE == 1) acc += lds[a[q]];
            if (MODE == 2) atomicOr(&lds[a[q]], 1u << (r & 31));
            if (MODE == 3) ((u64 *)lds)[a[q] >> 1] = (u64)r;
        }
    }
    const u64 t1 = clock64();
    if (acc == 0x12345u) out[1] = acc;
    __syncthreads();
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    u64 *out; hipMalloc(&out, 64);
    const int iters = 1000;
    const char *names[] = {"ds_or_rtn_b32 random", "ds_read_b32 random", "ds_or_b32 (no return) random", "ds_write_b64 random"};
#define RUN(M) { hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
    hipLaunchKernelGGL(k<M>, dim3(256), dim3(1024), 131072, 0, out, iters); hipDeviceSynchronize(); \
    u64 h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); \
    printf("%-32s %6.2f clk per wave-instruction per CU  (%5.2f lanes/clk)\n", names[M], (double)h / (iters * 8.0 * 16.0), 64.0 * iters * 8 * 16 / (double)h); }
    RUN(0) RUN(1) RUN(2) RUN(3)
    return 0;
}
