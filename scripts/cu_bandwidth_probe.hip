// Probe (gfx950): how many bytes per clock can ONE CU pull with 16 waves of coalesced 16-byte-per-lane loads of random
// 1 KiB chunks (the sweeps' access pattern), as a function of how many CUs do it at the same time and how many loads
// each wave keeps in flight?  Tells a per-CU request cap from the chip-wide HBM cap.
// build: hipcc --offload-arch=gfx950 -O3 scripts/cu_bandwidth_probe.hip -o /tmp/cubw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(1024) void k(const u32x4 *buf, u64 n_chunks, int iters, unsigned *sink, u64 *cycles, int misalign_dwords) {
    const int lane = threadIdx.x & 63;
    unsigned r = (blockIdx.x * 1024 + threadIdx.x / 64 * 64) * 2654435761u + 12345u;   // same per wave
    unsigned acc = 0;
    const u64 t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            r = r * 1664525u + 1013904223u;
            const u64 chunk = ((u64)r * n_chunks) >> 32;
            v[d] = *(const u32x4 *)((const unsigned *)(buf + chunk * 64 + lane) + misalign_dwords);     // 0 = 16-byte aligned, 1..3 = the sweeps' usual case
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
    }
    const u64 t1 = clock64();
    if (acc == 0x1234567u) sink[0] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main(int argc, char **argv) {
    const int mis = argc > 1 ? atoi(argv[1]) : 0;
    const u64 bytes = 1ull << 30;
    u32x4 *buf; unsigned *sink; u64 *cyc;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&sink, 64); hipMalloc(&cyc, 8 * 256);
    const u64 n_chunks = bytes / 1024 - 1;
    printf("misalignment: %d dwords\n", mis);
    const int grids[] = {8, 32, 128, 256};
    for (int depth = 2; depth <= 8; depth *= 2) {
        for (int g : grids) {
            const int iters = 4096 / depth;
            if (depth == 2) hipLaunchKernelGGL(k<2>, dim3(g), dim3(1024), 0, 0, buf, n_chunks, iters, sink, cyc, mis);
            if (depth == 4) hipLaunchKernelGGL(k<4>, dim3(g), dim3(1024), 0, 0, buf, n_chunks, iters, sink, cyc, mis);
            if (depth == 8) hipLaunchKernelGGL(k<8>, dim3(g), dim3(1024), 0, 0, buf, n_chunks, iters, sink, cyc, mis);
            hipDeviceSynchronize();
            u64 h[256]; hipMemcpy(h, cyc, 8 * g, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < g; ++i) avg += (double)h[i]; avg /= g;
            const double bytes_per_cu = 16.0 * 4096 * 1024;   // 16 waves x 4096 loads x 1 KiB
            printf("depth %d in flight per wave, %3d CUs busy: %6.2f B/clk/CU  (%.2f TB/s aggregate at 2.4 GHz)\n", depth, g, bytes_per_cu / avg,
                   bytes_per_cu / avg * g * 2.4e9 / 1e12);
        }
    }
    return 0;
}
