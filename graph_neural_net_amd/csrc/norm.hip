// GraphNorm statistics / apply kernels, error plumbing and tiny reductions (gfx950).
// Replaces the reductions and elementwise chain of models/layers.py:68-80
// (torch.mean, torch.var(unbiased=False), (b-mean)/(2*sqrt(n*(var+eps))), weight*.+bias)
// and their ragged variants (maskedtensors/maskedtensor.py:310-335, layers.py:79).
#include <stdarg.h>
#include <stdio.h>
#include "fgnn_common.h"

static thread_local char g_err[512] = "";

void fgnn_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *fgnn_last_error(void) { return g_err; }
extern "C" int fgnn_version(void) { return 1; }

namespace {

DEVI void write_nrm(float *nrm, long long idx, float mean, float m2, float m, float nv, float w, float eps) {
    const float var = m > 0.f ? m2 / m : 0.f;
    const float ve = var + eps;
    const float q = 1.f / (2.f * sqrtf(nv * ve));
    float4 o;
    o.x = mean;
    o.y = w * q;
    o.z = q;
    o.w = 1.f / ve;
    reinterpret_cast<float4 *>(nrm)[idx] = o;
}

// one thread per (g,c): Chan's pairwise update over the tiles, fixed order.
__global__ void gn_finalize_kernel(const float *part, const float *cnt, const float *gw, const int *nvalid,
                                   int G, int C, int N, int tpg, float eps, float *nrm) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * C) return;
    const int g = idx / C, c = idx - g * C;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int t = 0; t < tpg; ++t) {
        const float nb = cnt[(long long)g * tpg + t];
        if (nb > 0.f) {
            const float2 pm = reinterpret_cast<const float2 *>(part)[((long long)g * tpg + t) * C + c];
            const float delta = pm.x - mean;
            const float nn = n + nb;
            mean += delta * (nb / nn);
            m2 += pm.y + delta * delta * (n * nb / nn);
            n = nn;
        }
    }
    write_nrm(nrm, idx, mean, m2, n, (float)nvalid_of(nvalid, g, N), gw ? gw[c] : 1.f, eps);
}

// one wave per (g,c): two-pass mean / M2 over the valid n x n region.
__global__ void gn_stats_kernel(const float *x, long long gstride, long long ldp, const float *gw,
                                const int *nvalid, int G, int C, int N, float eps, float *nrm) {
    const int idx = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (idx >= G * C) return;
    const int lane = threadIdx.x & 63;
    const int g = idx / C, c = idx - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    const float *xp = x + (long long)g * gstride + (long long)c * ldp;
    const int P = N * N;
    float s = 0.f;
    for (int p = lane; p < P; p += WAVE) {
        const int i = p / N, j = p - i * N;
        if (i < nv && j < nv) s += xp[p];
    }
    const float m = (float)nv * (float)nv;
    const float mean = m > 0.f ? wave_sum(s) / m : 0.f;
    float s2 = 0.f;
    for (int p = lane; p < P; p += WAVE) {
        const int i = p / N, j = p - i * N;
        if (i < nv && j < nv) {
            const float d = xp[p] - mean;
            s2 += d * d;
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) write_nrm(nrm, idx, mean, s2, m, (float)nv, gw ? gw[c] : 1.f, eps);
}

__global__ void gn_apply_kernel(const float *z, long long zg, long long ldz, const float *nrm, const float *beta,
                                const int *nvalid, int C, int N, float *y, long long yg, long long ldy) {
    const int gc = blockIdx.y;
    const int g = gc / C, c = gc - g * C;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N * N) return;
    const int nv = nvalid_of(nvalid, g, N);
    const int i = p / N, j = p - i * N;
    const float4 n = reinterpret_cast<const float4 *>(nrm)[gc];
    const float v = z[(long long)g * zg + (long long)c * ldz + p];
    const float be = beta ? beta[c] : 0.f;
    y[(long long)g * yg + (long long)c * ldy + p] = (i < nv && j < nv) ? (v - n.x) * n.y + be : 0.f;
}

// S1 = sum dy, S2 = sum dy*(z-mean) per (g,c); one wave each.
__global__ void gn_bwd_stats_kernel(const float *dy, long long dg, long long ldd, const float *z, long long zg,
                                    long long ldz, const float *nrm, const int *nvalid, int G, int C, int N, float *s12) {
    const int idx = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (idx >= G * C) return;
    const int lane = threadIdx.x & 63;
    const int g = idx / C, c = idx - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    const float mean = nrm[(long long)idx * 4];
    const float *dp = dy + (long long)g * dg + (long long)c * ldd;
    const float *zp = z + (long long)g * zg + (long long)c * ldz;
    float s1 = 0.f, s2 = 0.f;
    const int P = N * N;
    for (int p = lane; p < P; p += WAVE) {
        const int i = p / N, j = p - i * N;
        if (i < nv && j < nv) {
            const float d = dp[p];
            s1 += d;
            s2 += d * (zp[p] - mean);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        s12[(long long)idx * 2] = s1;
        s12[(long long)idx * 2 + 1] = s2;
    }
}

// coef[g,c] = {mean, ca, cb, cc}: dz = ca*dy + cb*(z-mean) + cc  
__global__ void gn_bwd_coef_kernel(const float *s12, const float *nrm, const int *nvalid, int G, int C, int N, float *coef) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * C) return;
    const int g = idx / C;
    const float nv = (float)nvalid_of(nvalid, g, N);
    const float m = nv * nv;
    const float4 n = reinterpret_cast<const float4 *>(nrm)[idx];
    const float s1 = s12[(long long)idx * 2], s2 = s12[(long long)idx * 2 + 1];
    float4 o;
    o.x = n.x;
    o.y = n.y;
    o.z = m > 0.f ? -n.y * s2 * n.w / m : 0.f;
    o.w = m > 0.f ? -n.y * s1 / m : 0.f;
    reinterpret_cast<float4 *>(coef)[idx] = o;
}

// d gn_weight[c] = sum_g q[g,c]*S2[g,c], d gn_bias[c] = sum_g S1[g,c]; fixed order over g.
__global__ void gn_bwd_affine_kernel(const float *s12, const float *nrm, int G, int C, float *dgw, float *dgb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float aw = 0.f, ab = 0.f;
    for (int g = 0; g < G; ++g) {
        const long long idx = (long long)g * C + c;
        aw += nrm[idx * 4 + 2] * s12[idx * 2 + 1];
        ab += s12[idx * 2];
    }
    if (dgw) dgw[c] = aw;
    if (dgb) dgb[c] = ab;
}

__global__ void gn_bwd_apply_kernel(const float *dy, long long dg, long long ldd, const float *z, long long zg,
                                    long long ldz, const float *coef, const int *nvalid, int C, int N, float *dz,
                                    long long og, long long ldo) {
    const int gc = blockIdx.y;
    const int g = gc / C, c = gc - g * C;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N * N) return;
    const int nv = nvalid_of(nvalid, g, N);
    const int i = p / N, j = p - i * N;
    const float4 k = reinterpret_cast<const float4 *>(coef)[gc];
    const float d = dy[(long long)g * dg + (long long)c * ldd + p];
    const float u = z[(long long)g * zg + (long long)c * ldz + p] - k.x;
    dz[(long long)g * og + (long long)c * ldo + p] = (i < nv && j < nv) ? k.y * d + k.z * u + k.w : 0.f;
}

__global__ void reduce_rows_kernel(const float *in, int rows, int cols, float scale, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols) return;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += in[(long long)r * cols + i];
    out[i] = a * scale;
}

}  // namespace

extern "C" int fgnn_gn_finalize(const float *part, const float *cnt, const float *gn_weight, const int *nvalid,
                                int G, int C, int N, float eps, float *nrm, void *stream) {
    FGNN_CHECK(part && cnt && nrm && G > 0 && C > 0 && N > 0, "fgnn_gn_finalize: bad arguments");
    const int tot = G * C;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, cnt,
                       gn_weight, nvalid, G, C, N, fgnn_tiles_per_graph(N), eps, nrm);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_gn_stats(const float *x, long long gstride, long long ldp, const float *gn_weight,
                             const int *nvalid, int G, int C, int N, float eps, float *nrm, void *stream) {
    FGNN_CHECK(x && nrm && G > 0 && C > 0 && N > 0 && ldp >= (long long)N * N, "fgnn_gn_stats: bad arguments");
    const int tot = G * C;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((tot + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gstride, ldp,
                       gn_weight, nvalid, G, C, N, eps, nrm);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_gn_apply(const float *z, long long zgstride, long long ldz, const float *nrm, const float *beta,
                             const int *nvalid, int G, int C, int N, float *y, long long ygstride, long long ldy,
                             void *stream) {
    FGNN_CHECK(z && nrm && y && G > 0 && C > 0 && N > 0, "fgnn_gn_apply: bad arguments");
    FGNN_CHECK((long long)G * C < 65536ll * 32768, "fgnn_gn_apply: G*C too large");
    hipLaunchKernelGGL(gn_apply_kernel, dim3((N * N + 255) / 256, G * C), dim3(256), 0, (hipStream_t)stream, z,
                       zgstride, ldz, nrm, beta, nvalid, C, N, y, ygstride, ldy);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_gn_bwd_stats(const float *dy, long long dgstride, long long ldd, const float *z,
                                 long long zgstride, long long ldz, const float *nrm, const int *nvalid, int G, int C,
                                 int N, float *s12, void *stream) {
    FGNN_CHECK(dy && z && nrm && s12 && G > 0 && C > 0 && N > 0, "fgnn_gn_bwd_stats: bad arguments");
    const int tot = G * C;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3((tot + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, dgstride,
                       ldd, z, zgstride, ldz, nrm, nvalid, G, C, N, s12);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_gn_bwd_coef(const float *s12, const float *nrm, const int *nvalid, int G, int C, int N,
                                float *coef, float *dgn_w, float *dgn_b, void *stream) {
    FGNN_CHECK(s12 && nrm && coef && G > 0 && C > 0 && N > 0, "fgnn_gn_bwd_coef: bad arguments");
    const int tot = G * C;
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, s12, nrm,
                       nvalid, G, C, N, coef);
    FGNN_LAUNCH_CHECK();
    if (dgn_w || dgn_b) {
        hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, s12, nrm, G, C,
                           dgn_w, dgn_b);
        FGNN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int fgnn_gn_bwd_apply(const float *dy, long long dgstride, long long ldd, const float *z,
                                 long long zgstride, long long ldz, const float *coef, const int *nvalid, int G, int C,
                                 int N, float *dz, long long ogstride, long long ldo, void *stream) {
    FGNN_CHECK(dy && z && coef && dz && G > 0 && C > 0 && N > 0, "fgnn_gn_bwd_apply: bad arguments");
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((N * N + 255) / 256, G * C), dim3(256), 0, (hipStream_t)stream, dy,
                       dgstride, ldd, z, zgstride, ldz, coef, nvalid, C, N, dz, ogstride, ldo);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_reduce_partials(const float *wpart, int num_wg, int count, float *out, void *stream) {
    FGNN_CHECK(wpart && out && num_wg > 0 && count > 0, "fgnn_reduce_partials: bad arguments");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, wpart, num_wg,
                       count, 1.0f, out);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_sum_scale(const float *in, int rows, int cols, float scale, float *out, void *stream) {
    FGNN_CHECK(in && out && rows > 0 && cols > 0, "fgnn_sum_scale: bad arguments");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, rows, cols,
                       scale, out);
    FGNN_LAUNCH_CHECK();
    return 0;
}
