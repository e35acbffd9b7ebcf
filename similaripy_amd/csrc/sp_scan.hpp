// sp_scan.hpp — prefix sums of int32 counters (row pointer arrays) by many workgroups.
//
// The array is cut into SCAN_CHUNKS contiguous chunks.  Launch 1 sums every chunk (coalesced reads), launch 2 lets the
// workgroup of chunk j add up the chunk sums before it and scan its own chunk tile by tile.  8 KB of scratch, two short
// launches, instead of one workgroup walking the whole array.  Sums are carried in 64 bits; the outputs are the int32 row
// pointers of a CSR (the callers have checked that the total fits).
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr int SCAN_CHUNKS = 1024;
constexpr size_t SCAN_SCRATCH_BYTES = SCAN_CHUNKS * sizeof(long long);

__device__ __forceinline__ long long scan_wave_sum(long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// sum over the 256 threads of the workgroup, returned to all of them (sh: 4 slots)
__device__ __forceinline__ long long scan_block_sum(long long v, long long *sh) {
    v = scan_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void sp_scan_chunk_sums_kernel(long long n, const int *__restrict__ in, long long *__restrict__ part) {
    __shared__ long long sh[4];
    const long long per = (n + SCAN_CHUNKS - 1) / SCAN_CHUNKS;
    const long long b = min(n, blockIdx.x * per), e = min(n, b + per);
    long long s = 0;
    for (long long i = b + threadIdx.x; i < e; i += 256) s += in[i];
    s = scan_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// INCLUSIVE: out[i] = in[0] + ... + in[i]; else out[i] = in[0] + ... + in[i-1] and out[n] = the total.
// out may be `in` (every element is read by the thread that writes it, before it writes); out2 (may be NULL) gets a copy.
template <bool INCLUSIVE>
__global__ __launch_bounds__(256) void sp_scan_chunks_kernel(long long n, const int *in, int *out, int *out2, const long long *__restrict__ part, long long *total) {
    __shared__ long long sh[4];
    __shared__ long long wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long carry = 0;
    for (int i = tid; i < (int)blockIdx.x; i += 256) carry += part[i];
    carry = scan_block_sum(carry, sh);
    const long long per = (n + SCAN_CHUNKS - 1) / SCAN_CHUNKS;
    const long long b = min(n, blockIdx.x * per), e = min(n, b + per);
    for (long long t = b; t < e; t += 1024) {
        const long long i0 = t + tid * 4;
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i0 + j < e) ? in[i0 + j] : 0;
        const long long mine = (long long)v[0] + v[1] + v[2] + v[3];
        long long inc = mine;                                   // inclusive scan over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        __syncthreads();                                        // (wsum of the previous tile has been read)
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        long long run = carry + before + inc - mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < e) {
                if (INCLUSIVE) { run += v[j]; out[i0 + j] = (int)run; if (out2) out2[i0 + j] = (int)run; }
                else           { out[i0 + j] = (int)run; if (out2) out2[i0 + j] = (int)run; run += v[j]; }
            }
        }
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (blockIdx.x == SCAN_CHUNKS - 1 && tid == 0) {           // (its carry ends as the grand total, whatever its chunk held)
        if (!INCLUSIVE) out[n] = (int)carry;
        if (total) total[0] = carry;
    }
}

// exclusive: out[0..n] (n + 1 entries) from in[0..n); inclusive: out[0..n) in place or not.  scratch: SCAN_SCRATCH_BYTES.
template <bool INCLUSIVE>
inline void scan_i32(long long n, const int *in, int *out, int *out2, long long *total, void *scratch, hipStream_t stream) {
    long long *part = (long long *)scratch;
    hipLaunchKernelGGL(sp_scan_chunk_sums_kernel, dim3(SCAN_CHUNKS), dim3(256), 0, stream, n, in, part);
    hipLaunchKernelGGL((sp_scan_chunks_kernel<INCLUSIVE>), dim3(SCAN_CHUNKS), dim3(256), 0, stream, n, in, out, out2, (const long long *)part, total);
}

}  // namespace
