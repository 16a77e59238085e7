// Semantics probe of ds_read_b64_tr_b16 (gfx950): LDS is filled with the element index (16-bit), every lane supplies an address,
// and the four 16-bit results of every lane are printed.  build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_tr16 tools/ubench_tr16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(uint16_t *out, int stride_bytes, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned base = (unsigned)(uintptr_t)lds;
    unsigned addr;
    if (mode == 0) addr = base + stride_bytes * lane;                                   // lane-linear, stride_bytes apart
    else if (mode == 1) addr = base + stride_bytes * (lane & 15) + 8 * (lane >> 4);     // lane&15 -> row, lane>>4 -> 8-byte piece
    else addr = base + stride_bytes * (lane >> 4) + 8 * (lane & 15);                    // lane>>4 -> row, lane&15 -> 8-byte piece
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (uint16_t)(v[0] & 0xffff);
    out[lane * 4 + 1] = (uint16_t)(v[0] >> 16);
    out[lane * 4 + 2] = (uint16_t)(v[1] & 0xffff);
    out[lane * 4 + 3] = (uint16_t)(v[1] >> 16);
}

int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, sizeof(h));
    const int cfgs[][2] = {{8, 0}, {128, 1}, {32, 2}, {128, 2}};
    for (auto &c : cfgs) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d stride %d bytes (element index = byte/2):\n", c[1], c[0]);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
