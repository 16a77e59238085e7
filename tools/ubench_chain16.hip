// Is a conv chain on v_mfma_f32_16x16x4_f32 in the fragment layout of csrc/fgnn_t16.h bit-identical to the 32-pixel kernels' chain on
// v_mfma_f32_32x32x2_f32 (same bias start, same order of fused multiply-adds)?  One wave, random data, two layers with ReLU between.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_chain16 tools/ubench_chain16.hip && tools/ubench_chain16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __host__ constexpr int ch_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __host__ constexpr int chan(int s, int q) { return 8 * (s >> 1) + 2 * (s & 1) + (q >> 1) + 4 * (q & 1); }

// X (32 ch, 32 px), W0 / W1 (32 out, 32 in), b0 / b1; out (32 ch, 32 px) = W1 relu(W0 X + b0) + b1
__global__ void chain32(const float *X, const float *W0, const float *b0, const float *W1, const float *b1, float *out) {
    const int l = threadIdx.x, j = l & 31, h = l >> 5;
    f32x16 acc;
    float x[16];
    for (int r = 0; r < 16; ++r) { x[r] = X[ch_of(r, h) * 32 + j]; acc[r] = b0[ch_of(r, h)]; }
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W0[j * 32 + ch_of(r, h)], x[r], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { x[r] = fmaxf(acc[r], 0.f); acc[r] = b1[ch_of(r, h)]; }
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W1[j * 32 + ch_of(r, h)], x[r], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[ch_of(r, h) * 32 + j] = acc[r];
}
__global__ void chain16(const float *X, const float *W0, const float *b0, const float *W1, const float *b1, float *out) {
    const int l = threadIdx.x, px = l & 15, q = l >> 4, m = l & 15, k = l >> 4;
    for (int half = 0; half < 2; ++half) {
        f32x4 acc[2];
        float x[8];
        for (int s = 0; s < 8; ++s) { x[s] = X[chan(s, q) * 32 + 16 * half + px]; acc[s >> 2][s & 3] = b0[chan(s, q)]; }
        for (int s = 0; s < 8; ++s)
            for (int b = 0; b < 2; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(W0[chan(4 * b + (m & 3), m >> 2) * 32 + chan(s, k)], x[s], acc[b], 0, 0, 0);
        for (int s = 0; s < 8; ++s) { x[s] = fmaxf(acc[s >> 2][s & 3], 0.f); acc[s >> 2][s & 3] = b1[chan(s, q)]; }
        for (int s = 0; s < 8; ++s)
            for (int b = 0; b < 2; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1[chan(4 * b + (m & 3), m >> 2) * 32 + chan(s, k)], x[s], acc[b], 0, 0, 0);
        for (int s = 0; s < 8; ++s) out[chan(s, q) * 32 + 16 * half + px] = acc[s >> 2][s & 3];
    }
}
int main() {
    std::vector<float> h(32 * 32 * 3 + 64), o32(1024), o16(1024);
    unsigned s = 777u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 22); }
    float *d, *a, *b;
    hipMalloc(&d, h.size() * 4); hipMalloc(&a, 4096); hipMalloc(&b, 4096);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const float *X = d, *W0 = d + 1024, *W1 = d + 2048, *b0 = d + 3072, *b1 = d + 3104;
    chain32<<<1, 64>>>(X, W0, b0, W1, b1, a);
    chain16<<<1, 64>>>(X, W0, b0, W1, b1, b);
    hipMemcpy(o32.data(), a, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(o16.data(), b, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    double mx = 0;
    for (int i = 0; i < 1024; ++i) { bad += std::memcmp(&o32[i], &o16[i], 4) != 0; mx = fmax(mx, fabs((double)o32[i] - o16[i])); }
    printf("two-layer chain, 32x32x2 vs 16x16x4 in the t16 layout: %d of 1024 outputs differ (max abs diff %.3e)\n", bad, mx);
    return bad != 0;
}
