// Probe (gfx950): what do the building blocks of a barrier-delimited dense phase cost for ONE 1024-thread
// workgroup per CU?  s_memtime deltas per iteration, lane 0 of wave 0, averaged over many iterations.
// build: hipcc --offload-arch=gfx950 -O3 scripts/phase_cost_probe.hip -o /tmp/phase_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
template <int MODE>
__global__ __launch_bounds__(1024) void k(u64 *out, const unsigned *g, int iters) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += 1024) lds[i] = 0;
    __syncthreads();
    unsigned acc = 0;
    unsigned r = tid * 2654435761u;
    const u64 t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        r = r * 1664525u + 1013904223u;
        if (MODE == 0) { __syncthreads(); }
        if (MODE == 1) { acc += lds[(r >> 17) & 32767]; __syncthreads(); }                           // 1 random LDS read + barrier
        if (MODE == 2) { unsigned a = lds[(r >> 17) & 32767]; acc += lds[(a + r) & 32767]; __syncthreads(); }   // 2 dependent reads
        if (MODE == 3) { acc += atomicOr(&lds[(r >> 17) & 32767], 1u << (r & 31)); __syncthreads(); }  // returning atomic
        if (MODE == 4) { u64 *p = (u64 *)lds + ((r >> 18) & 16383); acc += (unsigned)atomicCAS(p, 0ull, (u64)r); __syncthreads(); }
        if (MODE == 5) { acc += g[(r >> 8) & 0xFFFFF]; __syncthreads(); }                              // random global load (4 MB table: L2)
        if (MODE == 6) { if (tid == 0) acc += atomicAdd(&lds[0], 1u); __syncthreads(); }                // one lane atomic + barrier
        if (MODE == 7) {   // 100 dependent VALU instructions per wave, all 16 waves
#pragma unroll
            for (int q = 0; q < 100; ++q) acc = acc * 3u + r;
            __syncthreads();
        }
        if (MODE == 8) {   // same, only wave 0
            if (tid < 64) {
#pragma unroll
                for (int q = 0; q < 100; ++q) acc = acc * 3u + r;
            }
            __syncthreads();
        }
    }
    const u64 t1 = clock64();
    if (acc == 0x12345u) out[1] = acc;
    if (tid == 0 && blockIdx.x == 0) out[0] = (t1 - t0);
}
int main() {
    u64 *out; unsigned *g;
    hipMalloc(&out, 64); hipMalloc(&g, 4 << 20); hipMemset(g, 0, 4 << 20);
    const int iters = 2000;
    const char *names[] = {"barrier only", "1 random ds_read + barrier", "2 dependent ds_reads + barrier", "ds_or_rtn + barrier",
                           "ds_cmpst_rtn_b64 + barrier", "random global load (L2) + barrier", "lane-0 atomic + barrier",
                           "100 dependent VALU x16 waves + barrier", "100 dependent VALU x1 wave + barrier"};
#define RUN(M) { hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
    hipLaunchKernelGGL(k<M>, dim3(256), dim3(1024), 131072, 0, out, g, iters); hipDeviceSynchronize(); \
    u64 h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); printf("%-45s %8.1f clk/iter\n", names[M], (double)h / iters); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    return 0;
}
