// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS kernel's access pattern on gfx950
// (MI355X_MICROARCH.md §HBM: the 2x under-report is established for 16 B/lane streams; other widths must be
// calibrated on a known byte count).  Pattern A = what sp_knn_rows_kernel does: each wave reads 64 consecutive
// dwords (256 B) starting at a pseudo-random 4-byte-aligned position of a 2 GiB buffer (m2 row slices);
// pattern B = plain dword-per-lane streaming; pattern W = dword-per-lane streaming writes.
//   hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o /tmp/fcal
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- /tmp/fcal
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr size_t N = (size_t)1 << 29;   // 2^29 dwords = 2 GiB
__global__ void calib_random_rows(const int *a, int *out, int iters) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned s = wave * 2654435761u + 12345u;
    int acc = 0;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const size_t base = (size_t)(s >> 3) % (N - 64);       // arbitrary (not 256 B aligned) start
        acc += a[base + lane];
    }
    if (acc == 0x12345678) out[0] = acc;
}
__global__ void calib_stream(const int *a, int *out, size_t n) {
    int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 0x12345678) out[0] = acc;
}
__global__ void calib_write(int *a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (int)i;
}
int main() {
    int *a, *out;
    hipMalloc(&a, N * 4); hipMalloc(&out, 4);
    hipMemset(a, 1, N * 4);
    const int blocks = 2048, threads = 256, iters = 4096;
    hipLaunchKernelGGL(calib_random_rows, dim3(blocks), dim3(threads), 0, 0, a, out, iters);
    hipLaunchKernelGGL(calib_stream, dim3(blocks), dim3(threads), 0, 0, a, out, N);
    hipLaunchKernelGGL(calib_write, dim3(blocks), dim3(threads), 0, 0, a, N);
    hipDeviceSynchronize();
    printf("calib_random_rows bytes %.0f\n", (double)blocks * threads / 64 * iters * 256.0);
    printf("calib_stream bytes %.0f\n", (double)N * 4);
    printf("calib_write bytes %.0f\n", (double)N * 4);
    return 0;
}
