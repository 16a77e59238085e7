// gfx950: what does the STRUCTURE of a t16 tile loop cost on one SIMD?  Two waves per SIMD each run
//   repeat { [NL ds_read_b128 of A operands] [16 x v_mfma_f32_16x16x4_f32 on two alternating accumulators] [NV VALU on the result -> next B operand] [NW ds_write_b32] }
// and the time per MFMA per SIMD is compared with the 32 cycles of the matrix pipe.  STAG: the second wave of a SIMD starts half a block late.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_t16_loop tools/ubench_t16_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)

template <int NL, int NV, int NW, int STAG, int PRE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += 512) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x4 acc[2];
    float b[8];
    for (int i = 0; i < 8; ++i) b[i] = 1.f + lane * 1e-4f + i;
    const float4 *img = reinterpret_cast<const float4 *>(lds) + lane;
    float *st = lds + 8192 + wv * 640 + lane;
    if (STAG && wv >= 4) __builtin_amdgcn_s_sleep(STAG);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float4 w[4];
    if (PRE) { for (int u = 0; u < 4; ++u) w[u] = img[u * 64]; }
    for (int it = 0; it < iters; ++it) {
        float4 wn[4];
        if (!PRE) {
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = NL > u ? img[((it & 3) * 4 + u) * 64] : make_float4(b[0], b[1], b[2], b[3]);
        }
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].x, b[2 * u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].y, b[2 * u], acc[1], 0, 0, 0);
            if (PRE && u == 0) {        // next block's operands requested under this block's MFMAs
#pragma unroll
                for (int v = 0; v < 4; ++v) wn[v] = img[(((it + 1) & 3) * 4 + v) * 64];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].z, b[2 * u + 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].w, b[2 * u + 1], acc[1], 0, 0, 0);
        }
        SB;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v = acc[s >> 2][s & 3];
#pragma unroll
            for (int e = 0; e < NV / 8; ++e) v = fmaxf(v * 0.5f, 1e-3f);
            b[s] = NV ? v : b[s] + 1e-9f * acc[s >> 2][s & 3];
        }
#pragma unroll
        for (int s = 0; s < NW; ++s) st[s * 64] = b[s & 7];
        if (PRE) {
#pragma unroll
            for (int v = 0; v < 4; ++v) w[v] = wn[v];
        }
        SB;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + acc[0][0];
    if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

// the same work per block -- 16 MFMAs, 4 ds_read_b128 (next block's operands), NV VALU pairs (on the PREVIOUS block's results: independent of the
// running chain), 8 ds_write_b32 -- but hand-interleaved: after every MFMA a slice of the other instructions, order frozen with sched_barrier
template <int NV>
__global__ __launch_bounds__(512) void kil(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += 512) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x4 acc[2], prev[2];
    float b[8], nb[8];
    for (int i = 0; i < 8; ++i) b[i] = nb[i] = 1.f + lane * 1e-4f + i;
    prev[0] = prev[1] = f32x4{1.f, 2.f, 3.f, 4.f};
    const float4 *img = reinterpret_cast<const float4 *>(lds) + lane;
    float *st = lds + 8192 + wv * 640 + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float4 w[4], wn[4];
    for (int u = 0; u < 4; ++u) w[u] = img[u * 64];
    for (int it = 0; it < iters; ++it) {
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int u = k >> 2, e = k & 3;
            const float wa = e == 0 ? w[u].x : e == 1 ? w[u].y : e == 2 ? w[u].z : w[u].w;
            acc[k & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, b[2 * u + (e >> 1)], acc[k & 1], 0, 0, 0);
            SB;
            if (k < 4) wn[k] = img[(((it + 1) & 3) * 4 + k) * 64];                 // next block's operands
            if (k >= 4 && k < 12) {                                                  // VALU on the previous block's result
                float v = prev[(k - 4) >> 2][(k - 4) & 3];
#pragma unroll
                for (int q = 0; q < NV / 8; ++q) v = fmaxf(v * 0.5f, 1e-3f);
                nb[k - 4] = v;
            }
            if (k >= 8) st[(k - 8) * 64] = nb[k - 8];                               // staging writes
            SB;
        }
        prev[0] = acc[0];
        prev[1] = acc[1];
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = nb[s];
#pragma unroll
        for (int v = 0; v < 4; ++v) w[v] = wn[v];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += b[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + acc[0][0] + prev[1][2];
    if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}
template <int NV>
void run_il() {
    const int blocks = 256, iters = 2000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    kil<NV><<<blocks, 512>>>(out, cyc, iters);
    kil<NV><<<blocks, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double per = s / h.size() / (iters * 16.0);
    printf("INTERLEAVED  ds_read_b128 4  valu %2d  ds_write 8 : %6.1f cycles per MFMA per wave = %5.1f per MFMA on the pipe (floor 32)\n", NV, per, per / 2);
    hipFree(out); hipFree(cyc);
}

template <int NL, int NV, int NW, int STAG, int PRE>
void run() {
    const int blocks = 256, iters = 2000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    k<NL, NV, NW, STAG, PRE><<<blocks, 512>>>(out, cyc, iters);
    k<NL, NV, NW, STAG, PRE><<<blocks, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    const double per = s / h.size() / (iters * 16.0);
    printf("ds_read_b128 %d  valu %2d  ds_write %d  stagger %2d  prefetch %d : %6.1f cycles per MFMA per wave = %5.1f per MFMA on the pipe (floor 32)\n", NL, NV, NW, STAG, PRE,
           per, per / 2);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0, 0, 0, 0>();
    run<4, 0, 0, 0, 0>();
    run<4, 8, 0, 0, 0>();
    run<4, 16, 0, 0, 0>();
    run<4, 32, 0, 0, 0>();
    run<4, 16, 8, 0, 0>();
    run<4, 16, 8, 8, 0>();
    run<4, 32, 8, 0, 0>();
    run<4, 32, 8, 6, 0>();
    run<4, 16, 8, 0, 1>();
    run<4, 32, 8, 0, 1>();
    run<0, 16, 8, 0, 0>();
    run<0, 32, 0, 0, 0>();
    run_il<16>();
    run_il<32>();
    run_il<48>();
    return 0;
}
