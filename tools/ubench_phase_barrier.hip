// What does a phase boundary cost inside a persistent kernel, against a kernel boundary in a replayed hipGraph?
// ("per-graph-synchronised multi-phase block backward".)
//
// Geometry of the headline workload: G = 64 graphs, one fp32 tensor = G x 80 000 floats (20.5 MB), 256 workgroups of 512
// threads with 128 KB of LDS (one per CU), i.e. four workgroups per graph.  A phase reads two tensors and writes one
// (the traffic of the forward matmul), after a prologue that copies a 32 KB "operand image" into LDS.  Workgroup q of
// graph g handles quarter (q + phase) % 4, so every phase reads what ANOTHER workgroup of the same graph wrote in the
// phase before: the dependency is per graph, never across graphs.
//   V0  P kernels in one hipGraph (what the engine does today)
//   V1  ONE kernel, per-graph arrival counters (4 workgroups), the graph's workgroups on one XCD (wg = xcd + 8 * slot)
//   V2  the same with the graph's workgroups spread over four XCDs (g = wg / 4)
//   V3  ONE kernel, device-wide arrival counter (256 workgroups) between phases
// Every wait is bounded (a failed wait sets an error flag and carries on): no hang if co-residency is not what we assume.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_phase_barrier tools/ubench_phase_barrier.hip && tools/ubench_phase_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int G = 64, PER_G = 80000, Q = PER_G / 4, NT = 512, IMG = 8192 /* floats */, WPG = 4;

struct Args {
    float *t[3];            // tensors, rotated per phase: phase p reads t[p % 3], t[(p + 1) % 3] and writes t[(p + 2) % 3]
    const float *img;       // [P][IMG]
    int *ctr;               // [P][G] (V1 / V2) or [P] (V3)
    int *err;
    int work;               // fmas per element
};

__device__ __forceinline__ void phase_body(const Args &A, int p, int g, int q, const float *imgl) {
    const float *a = A.t[p % 3] + (long long)g * PER_G, *b = A.t[(p + 1) % 3] + (long long)g * PER_G;
    float *o = A.t[(p + 2) % 3] + (long long)g * PER_G;
    const int rq = (q + p) & 3;
    const float w = imgl[threadIdx.x & (IMG - 1)];
    for (int e = rq * Q + threadIdx.x * 4; e < (rq + 1) * Q; e += NT * 4) {
        // a: the quarter that ANOTHER workgroup wrote in the previous phase (it wrote quarter (q' + p - 1) & 3)
        const float4 x = *reinterpret_cast<const float4 *>(a + e), y = *reinterpret_cast<const float4 *>(b + e);
        float4 r = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        for (int k = 0; k < A.work; ++k) {
            r.x = r.x * 0.999f + w; r.y = r.y * 0.999f + w; r.z = r.z * 0.999f + w; r.w = r.w * 0.999f + w;
        }
        *reinterpret_cast<float4 *>(o + e) = r;
    }
}
__device__ __forceinline__ void image_to_lds(float *l, const float *img) {
    for (int e = threadIdx.x * 4; e < IMG; e += NT * 4) *reinterpret_cast<float4 *>(l + e) = *reinterpret_cast<const float4 *>(img + e);
}

__global__ __launch_bounds__(NT) void one_phase(Args A, int p) {
    extern __shared__ float lds[];
    const int wg = blockIdx.x, xcd = wg & 7, slot = wg >> 3, g = xcd + 8 * (slot >> 2), q = slot & 3;
    image_to_lds(lds, A.img + (long long)p * IMG);
    __syncthreads();
    phase_body(A, p, g, q, lds);
}

// FENCE 0: every thread fences at agent scope on both sides (what a naive port writes);
// FENCE 1: __syncthreads() (workgroup-scope release: every wave's stores have reached the L2), then ONE thread releases /
//          acquires at agent scope (one L2 write-back + one invalidate per workgroup);
// FENCE 2: no agent-scope cache maintenance at all (relaxed counter; only valid if producers and consumers share an L2 and
//          consumers' L1 holds no stale line -- measures the floor, the checksum says whether it happened to work)
template <int FENCE>
__device__ __forceinline__ void arrive_wait(int *c, int target, int *err) {
    if (FENCE == 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE == 2) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(c, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int n = 0;
        while ((FENCE == 2 ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : __hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++n > (1 << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
    if (FENCE == 0) __threadfence();
}

template <int MODE, int FENCE>     // MODE 1: same XCD, 2: spread, 3: device-wide
__global__ __launch_bounds__(NT) void all_phases(Args A, int P) {
    extern __shared__ float lds[];
    const int wg = blockIdx.x;
    int g, q;
    if (MODE == 2) { g = wg >> 2; q = wg & 3; }
    else { const int xcd = wg & 7, slot = wg >> 3; g = xcd + 8 * (slot >> 2); q = slot & 3; }
    image_to_lds(lds, A.img);
    __syncthreads();
    for (int p = 0; p < P; ++p) {
        float *cur = lds + (p & 1) * IMG, *nxt = lds + ((p + 1) & 1) * IMG;
        phase_body(A, p, g, q, cur);
        if (p + 1 < P) {
            image_to_lds(nxt, A.img + (long long)(p + 1) * IMG);           // next phase's image: independent of the barrier
            if (MODE == 3) arrive_wait<FENCE>(A.ctr + p, gridDim.x, A.err);
            else arrive_wait<FENCE>(A.ctr + p * G + g, WPG, A.err);
        }
    }
}

int main(int argc, char **argv) {
    const int P = 8, lds = 128 * 1024, reps = 20;
    std::vector<float> h((size_t)G * PER_G, 1.f);
    Args A;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&A.t[i], h.size() * 4)); CK(hipMemcpy(A.t[i], h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    float *img; CK(hipMalloc(&img, (size_t)P * IMG * 4)); CK(hipMemset(img, 0, (size_t)P * IMG * 4)); A.img = img;
    CK(hipMalloc(&A.ctr, P * G * 4)); CK(hipMalloc(&A.err, 4)); CK(hipMemset(A.err, 0, 4));
    CK(hipFuncSetAttribute((const void *)one_phase, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    typedef void (*kern_t)(Args, int);
    struct Var { const char *name; kern_t k; };
    const Var vars[] = {{"V0 one kernel per phase (graph)", nullptr},
                        {"V1 per-graph counter, one XCD, all-thread fences", all_phases<1, 0>},
                        {"V1 per-graph counter, one XCD, one-thread fences", all_phases<1, 1>},
                        {"V1 per-graph counter, one XCD, NO cache maintenance", all_phases<1, 2>},
                        {"V2 per-graph counter, 4 XCDs, one-thread fences", all_phases<2, 1>},
                        {"V2 per-graph counter, 4 XCDs, NO cache maintenance", all_phases<2, 2>},
                        {"V3 device-wide counter, one-thread fences", all_phases<3, 1>}};
    const int NV = sizeof(vars) / sizeof(vars[0]);
    for (int v = 1; v < NV; ++v) CK(hipFuncSetAttribute((const void *)vars[v].k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.name, prop.multiProcessorCount);
    if (prop.multiProcessorCount < 256) { printf("needs 256 CUs for co-residency\n"); return 0; }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int work : {0, 16, 64}) {
        A.work = work;
        for (int v = 0; v < NV; ++v) {
            hipGraph_t gr; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            if (v == 0) {
                for (int p = 0; p < P; ++p) hipLaunchKernelGGL(one_phase, dim3(256), dim3(NT), lds, s, A, p);
            } else {
                CK(hipMemsetAsync(A.ctr, 0, P * G * 4, s));
                hipLaunchKernelGGL(vars[v].k, dim3(256), dim3(NT), lds, s, A, P);
            }
            CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            // one replay from a fresh state: the checksum must be the same in every variant (dependencies honoured)
            for (int i = 0; i < 3; ++i) CK(hipMemcpy(A.t[i], h.data(), h.size() * 4, hipMemcpyHostToDevice));
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            std::vector<float> back(h.size());
            double sum = 0;
            for (int i = 0; i < 3; ++i) {
                CK(hipMemcpy(back.data(), A.t[i], h.size() * 4, hipMemcpyDeviceToHost));
                for (float x : back) sum += x;
            }
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s)); for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int err; CK(hipMemcpy(&err, A.err, 4, hipMemcpyDeviceToHost));
            printf("work %3d  %-52s: %7.2f us per phase (61 MB each)  checksum %.6e%s\n", work, vars[v].name, ms * 1e3 / (reps * P), sum, err ? "  [WAIT TIMED OUT]" : "");
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        }
    }
    return 0;
}
