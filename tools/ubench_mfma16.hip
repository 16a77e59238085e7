// gfx950: v_mfma_f32_16x16x4_f32 -- operand / result layout and summation order, checked against an fmaf chain over k = 0..3 on the
// host-side model below (bit for bit), plus its issue rate next to v_mfma_f32_32x32x2_f32.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D (16x16) = C + A (16x4) B (4x16); one wave.  A[m][k], B[k][n] in global memory, D out.
__global__ void one(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16];          // lane l supplies A[m = l % 16][k = l / 16]
    const float b = B[(l / 16) * 16 + l % 16];         // lane l supplies B[k = l / 16][n = l % 16]
    f32x4 c;
    for (int v = 0; v < 4; ++v) c[v] = C[(4 * (l / 16) + v) * 16 + l % 16];      // lane l holds D[m = 4 (l / 16) + v][n = l % 16]
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + l % 16] = c[v];
}

template <int WIDE>
__global__ __launch_bounds__(512) void rate(float *out, unsigned long long *cyc, int iters) {
    float av = threadIdx.x * 1e-3f, bv = 1.f + threadIdx.x * 1e-4f;
    f32x16 w[2];
    f32x4 n[4];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) w[a][r] = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) n[a][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (WIDE) w[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, w[m & 1], 0, 0, 0);
            else {
                n[(2 * m) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, n[(2 * m) & 3], 0, 0, 0);
                n[(2 * m + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, n[(2 * m + 1) & 3], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += w[a][r];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) s += n[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    std::vector<float> A(64), B(64), C(256), D(256);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    for (auto &v : A) v = rnd();
    for (auto &v : B) v = rnd();
    for (auto &v : C) v = rnd();
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    one<<<1, 64>>>(dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad_fwd = 0, bad_rev = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            float f = C[m * 16 + n], r = C[m * 16 + n];
            for (int k = 0; k < 4; ++k) f = fmaf(A[m * 4 + k], B[k * 16 + n], f);
            for (int k = 3; k >= 0; --k) r = fmaf(A[m * 4 + k], B[k * 16 + n], r);
            bad_fwd += std::memcmp(&f, &D[m * 16 + n], 4) != 0;
            bad_rev += std::memcmp(&r, &D[m * 16 + n], 4) != 0;
        }
    printf("v_mfma_f32_16x16x4_f32 vs fmaf chain k = 0,1,2,3: %d of 256 elements differ; vs k = 3,2,1,0: %d differ\n", bad_fwd, bad_rev);
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int wide = 1; wide >= 0; --wide) {
        for (int rep = 0; rep < 2; ++rep) {
            if (wide) rate<1><<<256, 512>>>(out, cyc, 200); else rate<0><<<256, 512>>>(out, cyc, 200);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double t = 0; for (auto v : h) t += v;
        printf("%s, 2 waves per SIMD: %.1f s_memtime ticks per 4096 flop of a wave\n", wide ? "32x32x2 (one per 4096 flop)" : "16x16x4 (two per 4096 flop)",
               t / h.size() / (200 * 16.0));
    }
    return 0;
}
