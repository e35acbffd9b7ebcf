// Probe (gfx950): how does a raw buffer_load_dwordx4 behave when only PART of its 16 bytes lies inside
// num_records, and does it accept a 4-byte-aligned (not 16-byte-aligned) address?
// build: hipcc --offload-arch=gfx950 -O2 scripts/buffer_oob_probe.hip -o /tmp/buffer_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *p, unsigned nbytes, unsigned *out, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)nbytes, 0x00020000);
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
int main() {
    const int N = 4096;
    unsigned *d, *o, h[N], r[256];
    for (int i = 0; i < N; ++i) h[i] = 1000 + i;
    hipMalloc(&d, N * 4); hipMalloc(&o, 1024);
    hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
    // records end in the middle of lane 2's vector: valid dwords = 10 (+ soffset shift of 1 dword => misaligned by 4)
    for (int soff = 0; soff <= 4; soff += 4) {
        unsigned nbytes = 10 * 4 + soff;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nbytes, o, soff);
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("soffset=%d num_records=%u bytes (valid elements %d..%d):\n", soff, nbytes, soff / 4, 9 + soff / 4);
        for (int l = 0; l < 4; ++l) printf("  lane %d: %u %u %u %u\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    }
    return 0;
}
