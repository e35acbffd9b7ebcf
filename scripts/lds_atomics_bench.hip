// Microbenchmark: LDS atomic / plain op throughput on gfx950 with random (hash-like) addresses.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/lds_atomics_bench.hip -o /tmp/ldsb && /tmp/ldsb
// Reports lane-ops per cycle per CU for each op kind, at 4/8/16 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

constexpr int T = 16384;

enum Kind { ADD_F32_1LANE = -3, CAS64_1LANE = -2, ADD_F32_4LANE = -1, ADD_F32 = 0, CAS_RTN_B32, CAS_RTN_B64, ADD_RTN_U32, WRITE_B32, READ_B32, READ_B64, ADD_F32_SEQ, CAS_THEN_ADD, ADD_U64, CAS32_THEN_ADD_U64, KINDS };
static const char *names[KINDS] = {"ds_add_f32 (random)", "ds_cmpst_rtn_b32 (random)", "ds_cmpst_rtn_b64 (random)",
                                   "ds_add_rtn_u32 (random)", "ds_write_b32 (random)", "ds_read_b32 (random)",
                                   "ds_read_b64 (random)", "ds_add_f32 (lane-linear)", "cas_b32 + add_f32 (dependent)", "ds_add_u64 (random, no return)",
                                   "cas_b32 + add_u64 (dependent)"};

template <int KIND>
__global__ void k(int iters, unsigned long long *out, unsigned long long *cyc) {
    __shared__ unsigned long long tab[T];
    for (int i = threadIdx.x; i < T; i += blockDim.x) tab[i] = 0xFFFFFFFF00000000ull;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned long long acc = 0;
    float *fv = (float *)tab;
    int *iv = (int *)tab;
    unsigned *uv = (unsigned *)tab;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            unsigned slot = (s >> 10) & (T - 1);
            if (KIND == ADD_F32_1LANE) { if ((threadIdx.x & 63) == ((it + j) & 63)) atomicAdd(&fv[slot], 1.0f); }
            else if (KIND == ADD_F32_4LANE) { if ((threadIdx.x & 15) == ((it + j) & 15)) atomicAdd(&fv[slot], 1.0f); }
            else if (KIND == CAS64_1LANE) { if ((threadIdx.x & 63) == ((it + j) & 63)) acc += atomicCAS(&tab[slot], 0xFFFFFFFF00000000ull, (unsigned long long)slot); }
            else if (KIND == ADD_F32) atomicAdd(&fv[slot], 1.0f);
            else if (KIND == CAS_RTN_B32) acc += (unsigned)atomicCAS(&iv[slot], -1, (int)slot);
            else if (KIND == CAS_RTN_B64) acc += atomicCAS(&tab[slot], 0xFFFFFFFF00000000ull, (unsigned long long)slot);
            else if (KIND == ADD_RTN_U32) acc += atomicAdd(&uv[slot], 1u);
            else if (KIND == WRITE_B32) iv[slot] = (int)s;
            else if (KIND == READ_B32) acc += (unsigned)((volatile int *)iv)[slot];
            else if (KIND == READ_B64) acc += ((volatile unsigned long long *)tab)[slot];
            else if (KIND == ADD_F32_SEQ) atomicAdd(&fv[(threadIdx.x + it * 64 + j * 1024) & (T - 1)], 1.0f);
            else if (KIND == ADD_U64) atomicAdd(&tab[slot], (unsigned long long)s);
            else if (KIND == CAS32_THEN_ADD_U64) {
                int prev = atomicCAS(&iv[2 * (slot >> 1)], -1, (int)slot);
                if (prev == -1 || prev == (int)slot) atomicAdd(&tab[(T / 2) + (slot >> 1)], (unsigned long long)s);
            }
            else if (KIND == CAS_THEN_ADD) {
                int prev = atomicCAS(&iv[slot], -1, (int)slot);
                if (prev == -1 || prev == (int)slot) atomicAdd(&fv[T + slot], 1.0f);
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x1234567) out[0] = acc + tab[threadIdx.x & (T - 1)];
}

template <int KIND>
void run(int threads) {
    const int blocks = 256, iters = 2000;
    unsigned long long *out, *cyc;
    hipMalloc(&out, 8);
    hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, 10, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double ops = (double)threads * iters * 8;   // lane-ops per CU (1 block per CU)
    const char *nm = KIND == ADD_F32_1LANE ? "ds_add_f32, 1 active lane/wave" : KIND == ADD_F32_4LANE ? "ds_add_f32, 4 active lanes/wave" : KIND == CAS64_1LANE ? "ds_cmpst_rtn_b64, 1 active lane/wave" : names[KIND < 0 ? 0 : KIND];
    printf("%-38s %4d thr: %8.3f wave-instr-lanes/cycle/CU  (%.1f cycles per wave-instr, %.3f ms)\n", nm, threads, ops / avg,
           64.0 * avg / ops * (threads / 64) , ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<ADD_F32_1LANE>(threads); run<ADD_F32_4LANE>(threads); run<CAS64_1LANE>(threads);
        run<ADD_F32>(threads); run<CAS_RTN_B32>(threads); run<CAS_RTN_B64>(threads); run<ADD_RTN_U32>(threads);
        run<WRITE_B32>(threads); run<READ_B32>(threads); run<READ_B64>(threads); run<ADD_F32_SEQ>(threads); run<CAS_THEN_ADD>(threads); run<ADD_U64>(threads); run<CAS32_THEN_ADD_U64>(threads);
        printf("\n");
    }
    return 0;
}
