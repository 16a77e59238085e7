// Do two independent kernel chains overlap on this stack?  Each "kernel" is 256 workgroups of 256 threads with 72 KB of
// LDS (two fit a CU) that spin for ~20 us; a chain is 20 dependent launches.  Timed: one chain alone, two chains on two
// streams (eager), two chains captured fork/join into ONE hipGraph, and two chains as two graphs on two streams.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_concurrency tools/ubench_concurrency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void spin(float *out, long long ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    float acc = lds[(threadIdx.x + 1) & 255];
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < ticks) acc = acc * 1.0001f + 0.5f;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static int chain(hipStream_t s, float *buf, int n, int grid, int lds, long long ticks) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(grid), dim3(256), lds, s, buf, ticks);
    return 0;
}

int main() {
    const int n = 20, lds = 72 * 1024;
    const long long ticks = 50000;      // s_memtime ticks are shader cycles: ~22 us
    float *a, *b;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
    float ms;
    for (int grid : {256, 128}) {
        chain(s1, a, 5, grid, lds, ticks); CK(hipStreamSynchronize(s1));
        // one chain alone
        CK(hipEventRecord(e0, s1)); chain(s1, a, n, grid, lds, ticks); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %d: one chain of %d launches alone      : %.1f us per launch\n", grid, n, ms * 1e3 / n);
        // two chains, two streams, eager
        CK(hipEventRecord(e0, s1)); CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
        for (int i = 0; i < n; ++i) { chain(s1, a, 1, grid, lds, ticks); chain(s2, b, 1, grid, lds, ticks); }
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %d: two chains on two streams (eager)    : %.1f us per launch pair\n", grid, ms * 1e3 / n);
        // two chains captured fork/join into one graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
        chain(s1, a, n, grid, lds, ticks); chain(s2, b, n, grid, lds, ticks);
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
        CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventRecord(e0, s1)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s1)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %d: two chains, ONE graph (fork / join)  : %.1f us per launch pair\n", grid, ms * 1e3 / (5 * n));
        // one chain per graph, two graphs on two streams
        hipGraph_t g1, g2; hipGraphExec_t ge1, ge2;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal)); chain(s1, a, n, grid, lds, ticks); CK(hipStreamEndCapture(s1, &g1));
        CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal)); chain(s2, b, n, grid, lds, ticks); CK(hipStreamEndCapture(s2, &g2));
        CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge1, s1)); CK(hipGraphLaunch(ge2, s2)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1)); CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
        for (int r = 0; r < 5; ++r) { CK(hipGraphLaunch(ge1, s1)); CK(hipGraphLaunch(ge2, s2)); }
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %d: two graphs on two streams            : %.1f us per launch pair\n", grid, ms * 1e3 / (5 * n));
        // single graph with one chain (graph launch overhead reference)
        CK(hipEventRecord(e0, s1)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge1, s1)); CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %d: one chain as a graph                 : %.1f us per launch\n", grid, ms * 1e3 / (5 * n));
    }
    return 0;
}
