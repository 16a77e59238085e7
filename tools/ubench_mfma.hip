// Micro-benchmark (gfx950): how many filler instructions fit between v_mfma_f32_32x32x2_f32 issues,
// on the same / on alternating accumulators, with 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SB __builtin_amdgcn_sched_barrier(0)

template <int NACC, int KV, int KDSR, int KDSW>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ float lds[8 * 64 * 8];
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = threadIdx.x * 1e-3f, bv = 1.f + threadIdx.x * 1e-4f;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = av + i;
    float4 ld = make_float4(0, 0, 0, 0);
    float *my = lds + (threadIdx.x >> 6) * 512 + (threadIdx.x & 63) * 4;
    my[0] = av; my[1] = bv; my[2] = av; my[3] = bv;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m % NACC], 0, 0, 0);
            SB;
#pragma unroll
            for (int v = 0; v < KV; ++v) f[v % 8] = f[v % 8] * bv + av;
#pragma unroll
            for (int v = 0; v < KDSR; ++v) { const float4 t = reinterpret_cast<const float4 *>(lds)[((it + m + v) & 7) * 64 + (threadIdx.x & 63)]; ld.x += t.x; ld.y += t.w; }
#pragma unroll
            for (int v = 0; v < KDSW; ++v) lds[(threadIdx.x >> 6) * 512 + ((it + v) & 7) * 64 + (threadIdx.x & 63)] = f[v % 8];
            SB;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ld.x + ld.y;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int KV, int KDSR, int KDSW>
void run(int waves_per_simd) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 200;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8 * 8);
    k<NACC, KV, KDSR, KDSW><<<blocks, threads>>>(out, cyc, iters);
    k<NACC, KV, KDSR, KDSW><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double per = s / h.size() / (iters * 16.0);
    printf("acc=%d valu=%2d dsr=%d dsw=%d waves/simd=%d : %6.1f cyc per MFMA per wave  (pipe %.1f cyc/MFMA)\n", NACC, KV, KDSR, KDSW,
           waves_per_simd, per, per / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

// effective shader clock under a sustained pure-MFMA load: wall time of N dependent MFMAs at 64 cycles each
void clock_probe(int waves_per_simd) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 40000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<1, 0, 0, 0><<<blocks, threads>>>(out, cyc, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<1, 0, 0, 0><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double cycles = s / h.size();
    const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
    printf("clock probe, %d waves/SIMD: %.2f ms wall, %.0f s_memtime ticks per wave -> %.3f GHz tick rate; %.1f ticks per MFMA; "
           "fp32 MFMA rate %.1f TFLOP/s (256 CUs)\n", waves_per_simd, ms, cycles, cycles / (ms * 1e6), cycles / (iters * 16.0 * waves_per_simd),
           mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    clock_probe(1); clock_probe(2);
    for (int w = 1; w <= 2; ++w) {
        run<1, 0, 0, 0>(w); run<1, 1, 0, 0>(w); run<1, 4, 0, 0>(w); run<1, 8, 0, 0>(w); run<1, 12, 0, 0>(w); run<1, 16, 0, 0>(w);
        run<2, 0, 0, 0>(w); run<2, 1, 0, 0>(w); run<2, 4, 0, 0>(w); run<2, 8, 0, 0>(w); run<2, 12, 0, 0>(w); run<2, 16, 0, 0>(w);
        run<1, 0, 1, 0>(w); run<1, 0, 0, 1>(w); run<1, 4, 1, 1>(w); run<2, 4, 1, 1>(w); run<2, 8, 1, 2>(w); run<2, 4, 2, 4>(w);
    }
    return 0;
}
