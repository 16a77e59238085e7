// GraphNorm statistics / apply kernels, error plumbing and tiny reductions (gfx950).
// Replaces the reductions and elementwise chain of models/layers.py:68-80
// (torch.mean, torch.var(unbiased=False), (b-mean)/(2*sqrt(n*(var+eps))), weight*.+bias)
// and their ragged variants (maskedtensors/maskedtensor.py:310-335, layers.py:79).
#include <stdarg.h>
#include <stdio.h>
#include "fgnn_common.h"
#include "fgnn_pack.h"
#include "fgnn_norm.h"

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
    reinterpret_cast<float4 *>(nrm)[idx] = nrm_record(mean, m2, m, nv, w, eps);
}

// one wave per (g, c, MLP): exact two-level decomposition with plain (fixed-tree) wave sums,
//   mean = sum_t n_t mean_t / sum_t n_t,   M2 = sum_t [ M2_t + n_t (mean_t - mean)^2 ]
// (no cancellation: the between-tile term is formed from differences of means).
struct FinalizeJobs {
    const float *part[2];
    const float *gw[2];
    float *nrm[2];
};
__global__ __launch_bounds__(256) void gn_finalize_kernel(const FinalizeJobs J, const float *cnt, const int *nvalid,
                                                         int G, int C, int N, int tpg, float eps, int tr) {
    const float *part = J.part[blockIdx.y];
    const float *gw = J.gw[blockIdx.y];
    float *nrm = J.nrm[blockIdx.y];
    const int idx = xcd_swizzle(blockIdx.x, gridDim.x) * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (idx >= G * C) return;
    const int lane = threadIdx.x & 63;
    const int g = idx / C, c = idx - g * C;
    float sn = 0.f, sm = 0.f;
    const int nt = (tpg + WAVE - 1) / WAVE;
    if (nt <= 4) {       // up to 256 tiles per graph (N <= 90) in registers; else looped below
        const float4 r = finalize_wave(part, cnt, g, c, C, tpg, (float)nvalid_of(nvalid, g, N), gw ? gw[c] : 1.f, eps, lane, tr != 0);
        if (lane == 0) reinterpret_cast<float4 *>(nrm)[idx] = r;
    } else {
        // eight loads in flight per lane and pass (a plain loop pays one memory round trip per 64 tiles)
        constexpr int U = 8;
        const float *cg = cnt + (long long)g * tpg;
        const int ts = tr ? 1 : C;              // tile stride of the partials (float2 units), see fgnn_norm.h
        const float2 *pg = reinterpret_cast<const float2 *>(part) + (long long)g * tpg * C + (long long)c * (tr ? tpg : 1);
        for (int t0 = lane; t0 < tpg; t0 += U * WAVE) {
            float n[U], x[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int t = t0 + k * WAVE, tc = t < tpg ? t : 0;
                n[k] = t < tpg ? cg[tc] : 0.f;
                x[k] = pg[(long long)tc * ts].x;
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                sn += n[k];
                sm += n[k] * x[k];
            }
        }
        sn = wave_sum(sn);
        sm = wave_sum(sm);
        const float mean = sn > 0.f ? sm / sn : 0.f;
        float m2 = 0.f;
        for (int t0 = lane; t0 < tpg; t0 += U * WAVE) {
            float n[U];
            float2 pm[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int t = t0 + k * WAVE, tc = t < tpg ? t : 0;
                n[k] = t < tpg ? cg[tc] : 0.f;
                pm[k] = pg[(long long)tc * ts];
                if (t >= tpg) pm[k].y = 0.f;
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float d = pm[k].x - mean;
                m2 += pm[k].y + n[k] * d * d;
            }
        }
        m2 = wave_sum(m2);
        if (lane == 0) write_nrm(nrm, idx, mean, m2, sn, (float)nvalid_of(nvalid, g, N), gw ? gw[c] : 1.f, eps);
    }
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
    float s = 0.f, s2 = 0.f;
    const float m = (float)nv * (float)nv;
    float mean;
    if (nv == N) {          // dense plane: no index arithmetic, eight loads in flight per lane (same summation order)
        constexpr int U = 8;
        int p = lane;
        for (; p + (U - 1) * WAVE < P; p += U * WAVE) {
            float v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = xp[p + k * WAVE];
#pragma unroll
            for (int k = 0; k < U; ++k) s += v[k];
        }
        for (; p < P; p += WAVE) s += xp[p];
        mean = m > 0.f ? wave_sum(s) / m : 0.f;
        p = lane;
        for (; p + (U - 1) * WAVE < P; p += U * WAVE) {
            float v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = xp[p + k * WAVE];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const float d = v[k] - mean;
                s2 += d * d;
            }
        }
        for (; p < P; p += WAVE) {
            const float d = xp[p] - mean;
            s2 += d * d;
        }
    } else {                // ragged: walk the valid rows (the padding is never touched)
        for (int i = 0; i < nv; ++i)
            for (int j = lane; j < nv; j += WAVE) s += xp[i * N + j];
        mean = m > 0.f ? wave_sum(s) / m : 0.f;
        for (int i = 0; i < nv; ++i)
            for (int j = lane; j < nv; j += WAVE) {
                const float d = xp[i * N + j] - mean;
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
    bool ok = true;
    if (nv < N) {           // (uniform per workgroup: a dense plane needs no index arithmetic)
        const int i = p / N, j = p - i * N;
        ok = i < nv && j < nv;
    }
    const float4 n = reinterpret_cast<const float4 *>(nrm)[gc];
    const float v = z[(long long)g * zg + (long long)c * ldz + p];
    const float be = beta ? beta[c] : 0.f;
    y[(long long)g * yg + (long long)c * ldy + p] = ok ? (v - n.x) * n.y + be : 0.f;
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
    if (nv == N) {          // dense plane: same summation order, four pairs of loads in flight per lane
        constexpr int U = 4;
        int p = lane;
        for (; p + (U - 1) * WAVE < P; p += U * WAVE) {
            float d[U], z4[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                d[k] = dp[p + k * WAVE];
                z4[k] = zp[p + k * WAVE];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                s1 += d[k];
                s2 += d[k] * (z4[k] - mean);
            }
        }
        for (; p < P; p += WAVE) {
            const float d = dp[p];
            s1 += d;
            s2 += d * (zp[p] - mean);
        }
    } else {
        for (int i = 0; i < nv; ++i)
            for (int j = lane; j < nv; j += WAVE) {
                const float d = dp[i * N + j];
                s1 += d;
                s2 += d * (zp[i * N + j] - mean);
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
struct CoefJobs {
    const float *s12[2];
    const float *nrm[2];
    float *coef[2];
};
__global__ void gn_bwd_coef_kernel(const CoefJobs J, const int *nvalid, int G, int C, int N) {
    const float *s12 = J.s12[blockIdx.y];
    const float *nrm = J.nrm[blockIdx.y];
    float *coef = J.coef[blockIdx.y];
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

// d gn_weight[c] = sum_g q[g,c]*S2[g,c], d gn_bias[c] = sum_g S1[g,c]; fixed order.
// one block of 256 threads: thread (c = tid % 32 within a 32-channel chunk, gg = tid / 32) strides g.
DEVI void affine_reduce(const float *s12, const float *nrm, int G, int C, float *dgw, float *dgb, float *sm) {
    const int tid = threadIdx.x, cl = tid & 31, gg = tid >> 5;
    for (int c0 = 0; c0 < C; c0 += 32) {
        const int c = c0 + cl;
        float aw = 0.f, ab = 0.f;
        if (c < C)
            for (int g = gg; g < G; g += 8) {
                const long long idx = (long long)g * C + c;
                aw += nrm[idx * 4 + 2] * s12[idx * 2 + 1];
                ab += s12[idx * 2];
            }
        sm[tid] = aw;
        sm[256 + tid] = ab;
        __syncthreads();
        if (tid < 32 && c < C) {
            float w = 0.f, b = 0.f;
            for (int k = 0; k < 8; ++k) {
                w += sm[k * 32 + tid];
                b += sm[256 + k * 32 + tid];
            }
            if (dgw) dgw[c] = w;
            if (dgb) dgb[c] = b;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gn_bwd_affine_kernel(const float *s12, const float *nrm, int G, int C, float *dgw,
                                                            float *dgb) {
    __shared__ float sm[512];
    affine_reduce(s12, nrm, G, C, dgw, dgb, sm);
}

// sums per-tile {S1,S2} partials (G,tpg,C,2) over the tiles (one wave per (g,c), fixed tree),
// writes s12 (G*C*2) and the dz coefficients.
__global__ void gn_bwd_coef_tiles_kernel(const float *s12part, const float *nrm, const int *nvalid, int G, int C, int N,
                                         int tpg, float *s12, float *coef, int tr) {
    const int idx = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (idx >= G * C) return;
    const int lane = threadIdx.x & 63;
    const int g = idx / C, c = idx - g * C;
    float s1 = 0.f, s2 = 0.f;
    constexpr int U = 8;       // loads in flight per lane
    const int ts = tr ? 1 : C;
    const float2 *pg = reinterpret_cast<const float2 *>(s12part) + (long long)g * tpg * C + (long long)c * (tr ? tpg : 1);
    for (int t0 = lane; t0 < tpg; t0 += U * WAVE) {
        float2 p[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int t = t0 + k * WAVE;
            p[k] = pg[(long long)(t < tpg ? t : 0) * ts];
            if (t >= tpg) p[k] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            s1 += p[k].x;
            s2 += p[k].y;
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        const float nv = (float)nvalid_of(nvalid, g, N);
        const float m = nv * nv;
        const float4 n = reinterpret_cast<const float4 *>(nrm)[idx];
        s12[(long long)idx * 2] = s1;
        s12[(long long)idx * 2 + 1] = s2;
        float4 o;
        o.x = n.x;
        o.y = n.y;
        o.z = m > 0.f ? -n.y * s2 * n.w / m : 0.f;
        o.w = m > 0.f ? -n.y * s1 / m : 0.f;
        reinterpret_cast<float4 *>(coef)[idx] = o;
    }
}

__global__ void gn_bwd_apply_kernel(const float *dy, long long dg, long long ldd, const float *z, long long zg,
                                    long long ldz, const float *coef, const int *nvalid, int C, int N, float *dz,
                                    long long og, long long ldo) {
    const int gc = blockIdx.y;
    const int g = gc / C, c = gc - g * C;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N * N) return;
    const int nv = nvalid_of(nvalid, g, N);
    bool ok = true;
    if (nv < N) {
        const int i = p / N, j = p - i * N;
        ok = i < nv && j < nv;
    }
    const float4 k = reinterpret_cast<const float4 *>(coef)[gc];
    const float d = dy[(long long)g * dg + (long long)c * ldd + p];
    const float u = z[(long long)g * zg + (long long)c * ldz + p] - k.x;
    dz[(long long)g * og + (long long)c * ldo + p] = ok ? k.y * d + k.z * u + k.w : 0.f;
}

// out[i] = scale * sum_r in[r][i]; block = 4 row groups x 64 columns, fixed order.
DEVI void reduce_cols(const float *in, int rows, int cols, float scale, float *out, int cblock, float *sm) {
    const int tid = threadIdx.x, cl = tid & 63, rg = tid >> 6;
    const int i = cblock * 64 + cl;
    const int ic = i < cols ? i : 0;              // clamped: loads stay unconditional
    float a = 0.f;
    int r = rg;
    for (; r + 124 < rows; r += 128) {            // 32 independent loads in flight (256 workgroup rows: two round trips, not eight)
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = in[(long long)(r + 4 * k) * cols + ic];
#pragma unroll
        for (int k = 0; k < 32; ++k) a += v[k];
    }
    for (; r + 28 < rows; r += 32) {              // 8 independent loads in flight
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = in[(long long)(r + 4 * k) * cols + ic];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += v[k];
    }
    for (; r < rows; r += 4) a += in[(long long)r * cols + ic];
    sm[tid] = a;
    __syncthreads();
    if (tid < 64 && i < cols) out[i] = ((sm[tid] + sm[64 + tid]) + (sm[128 + tid] + sm[192 + tid])) * scale;
}

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float *in, int rows, int cols, float scale, float *out) {
    __shared__ float sm[256];
    reduce_cols(in, rows, cols, scale, out, blockIdx.x, sm);
}

struct GradJobs {
    fgnn_grad_job job[FGNN_MAX_GRAD_JOBS];
};

// grid (max column blocks + 1, njobs): column blocks reduce the workgroup partials of one MLP,
// the extra block reduces its GraphNorm affine gradients over the graphs.
__global__ __launch_bounds__(256) void grad_finalize_kernel(const GradJobs J, int num_wg, int G, int C) {
    __shared__ float sm[512];
    const fgnn_grad_job &jb = J.job[blockIdx.y];
    const int nblk = (jb.count + 63) / 64;
    if ((int)blockIdx.x < nblk) {
        reduce_cols(jb.wpart, jb.rows > 0 ? jb.rows : num_wg, jb.count, (jb.scale != 0.f ? jb.scale : 1.f) * (jb.scale_dev ? *jb.scale_dev : 1.f), jb.out,
                    blockIdx.x, sm);
    } else if ((int)blockIdx.x == gridDim.x - 1 && jb.s12) {
        affine_reduce(jb.s12, jb.nrm, G, C, jb.dgn_w, jb.dgn_b, sm);
    }
}

// grid (blocks per job, njobs): writes the LDS operand image(s) of one MLP kernel launch
__global__ __launch_bounds__(256) void pack_operands_kernel(const PackJobs J) {
    pack_job_body(J.job[blockIdx.y], blockIdx.x, gridDim.x, threadIdx.x);
}

}  // namespace

namespace {
__global__ __launch_bounds__(64) void inv_node_count_kernel(const int *nvalid, int B, float *out) {
    int s = 0;
    for (int b = threadIdx.x; b < B; b += 64) s += nvalid[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[0] = s > 0 ? 1.f / (float)s : 0.f;
}
}  // namespace

extern "C" int fgnn_inv_node_count(const int *nvalid, int B, float *out, void *stream) {
    FGNN_CHECK(nvalid && out && B > 0, "fgnn_inv_node_count: bad arguments");
    hipLaunchKernelGGL(inv_node_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nvalid, B, out);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_pack_floats(int kind, int ca, int cb, int depth, int nmlp) {
    if (kind >= 4)      // the *_t16 kernels' images: same layouts, each image padded to whole KiB (copied with global_load_lds, fgnn_pack.h)
        return kind == 4 ? pk_pad_floats(pk_fwd(ca, cb, depth).floats * nmlp) : pk_pad_floats(pk_bwd(ca, cb, depth).floats);
    return kind == 0 ? pk_fwd(ca, cb, depth).floats * nmlp : pk_bwd(ca, cb, depth).floats;
}

extern "C" int fgnn_pack_operands(const fgnn_pack_job *jobs, int njobs, void *stream) {
    FGNN_CHECK(jobs && njobs > 0 && njobs <= FGNN_MAX_PACK_JOBS, "fgnn_pack_operands: bad arguments (njobs=%d)", njobs);
    PackJobs J;
    for (int i = 0; i < njobs; ++i) {
        FGNN_CHECK(jobs[i].out && jobs[i].depth >= 1 && jobs[i].depth <= FGNN_MAX_DEPTH && (jobs[i].nmlp == 1 || jobs[i].nmlp == 2),
                   "fgnn_pack_operands: job %d malformed", i);
        J.job[i] = jobs[i];
    }
    hipLaunchKernelGGL(pack_operands_kernel, dim3(PACK_BLOCKS_PER_JOB, njobs), dim3(256), 0, (hipStream_t)stream, J);
    FGNN_LAUNCH_CHECK();
    return 0;
}

// tr = 1: partials stored (G, C, tpg, 2) (the bf16 kernels), 0: (G, tpg, C, 2) (the fp32 kernels)
static int gn_finalize_launch(const float *part, const float *cnt, const float *gn_weight, const int *nvalid, int G, int C, int N,
                              int tpg, float eps, float *nrm, int tr, void *stream) {
    FGNN_CHECK(part && cnt && nrm && G > 0 && C > 0 && N > 0 && tpg > 0, "fgnn_gn_finalize: bad arguments");
    const int tot = G * C;
    FinalizeJobs J = {{part, nullptr}, {gn_weight, nullptr}, {nrm, nullptr}};
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((tot + 3) / 4, 1), dim3(256), 0, (hipStream_t)stream, J, cnt,
                       nvalid, G, C, N, tpg, eps, tr);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_gn_finalize_tpg(const float *part, const float *cnt, const float *gn_weight, const int *nvalid,
                                    int G, int C, int N, int tpg, float eps, float *nrm, void *stream) {
    return gn_finalize_launch(part, cnt, gn_weight, nvalid, G, C, N, tpg, eps, nrm, 1, stream);
}
// _r: the tile statistics hold `recs` records per graph instead of fgnn_tiles_per_graph(N) (fgnn_mlp_fwd_t16: one per 16-pixel half)
extern "C" int fgnn_gn_finalize_r(const float *part, const float *cnt, const float *gn_weight, const int *nvalid,
                                  int G, int C, int N, int recs, float eps, float *nrm, void *stream) {
    return gn_finalize_launch(part, cnt, gn_weight, nvalid, G, C, N, recs, eps, nrm, 0, stream);
}
extern "C" int fgnn_gn_finalize(const float *part, const float *cnt, const float *gn_weight, const int *nvalid,
                                int G, int C, int N, float eps, float *nrm, void *stream) {
    return gn_finalize_launch(part, cnt, gn_weight, nvalid, G, C, N, fgnn_tiles_per_graph(N), eps, nrm, 0, stream);
}

static int gn_finalize2_launch(const float *part0, const float *part1, const float *cnt, const float *gn_weight0,
                               const float *gn_weight1, const int *nvalid, int G, int C, int N, int tpg, float eps, float *nrm0,
                               float *nrm1, int tr, void *stream) {
    FGNN_CHECK(part0 && part1 && cnt && nrm0 && nrm1 && G > 0 && C > 0 && N > 0 && tpg > 0, "fgnn_gn_finalize2: bad arguments");
    const int tot = G * C;
    FinalizeJobs J = {{part0, part1}, {gn_weight0, gn_weight1}, {nrm0, nrm1}};
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((tot + 3) / 4, 2), dim3(256), 0, (hipStream_t)stream, J, cnt,
                       nvalid, G, C, N, tpg, eps, tr);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_gn_finalize2_tpg(const float *part0, const float *part1, const float *cnt, const float *gn_weight0,
                                     const float *gn_weight1, const int *nvalid, int G, int C, int N, int tpg, float eps,
                                     float *nrm0, float *nrm1, void *stream) {
    return gn_finalize2_launch(part0, part1, cnt, gn_weight0, gn_weight1, nvalid, G, C, N, tpg, eps, nrm0, nrm1, 1, stream);
}
extern "C" int fgnn_gn_finalize2_r(const float *part0, const float *part1, const float *cnt, const float *gn_weight0,
                                   const float *gn_weight1, const int *nvalid, int G, int C, int N, int recs, float eps,
                                   float *nrm0, float *nrm1, void *stream) {
    return gn_finalize2_launch(part0, part1, cnt, gn_weight0, gn_weight1, nvalid, G, C, N, recs, eps, nrm0, nrm1, 0, stream);
}
extern "C" int fgnn_gn_finalize2(const float *part0, const float *part1, const float *cnt, const float *gn_weight0,
                                 const float *gn_weight1, const int *nvalid, int G, int C, int N, float eps,
                                 float *nrm0, float *nrm1, void *stream) {
    return gn_finalize2_launch(part0, part1, cnt, gn_weight0, gn_weight1, nvalid, G, C, N, fgnn_tiles_per_graph(N), eps, nrm0, nrm1,
                               0, stream);
}

// ---- GraphNorm of a dense (G, C, ld) tensor with ONE workgroup per (g, c) plane (N*N <= 4096): the plane is read once into
// registers, reduced in the workgroup (fixed order) and normalised from the registers -- statistics + apply in one pass over
// the tensor and one launch instead of two passes and two launches (three and three in the backward).
namespace {
constexpr int PLANE_EPT = 16, PLANE_THREADS = 256, PLANE_MAX = PLANE_EPT * PLANE_THREADS;

DEVI float plane_sum(float v, float *sm, int tid) {      // every thread gets the total; fixed order
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// validity of element p of the plane (dense planes need no index arithmetic)
DEVI bool plane_valid(int p, int P, int N, int nv) {
    if (p >= P) return false;
    if (nv == N) return true;
    const int i = p / N, j = p - i * N;
    return i < nv && j < nv;
}

__global__ __launch_bounds__(PLANE_THREADS) void gn_plane_fwd_kernel(const float *x, long long gs, long long ld, const float *gw,
                                                                     const float *beta, const int *nvalid, int C, int N, float eps,
                                                                     float *y, long long yg, long long ldy, float *nrm) {
    __shared__ float sm[4];
    const int gc = blockIdx.x, g = gc / C, c = gc - g * C, tid = threadIdx.x;
    const int nv = nvalid_of(nvalid, g, N), P = N * N;
    const float *xp = x + (long long)g * gs + (long long)c * ld;
    float v[PLANE_EPT];
    unsigned ok = 0;
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) {
        const int p = tid + PLANE_THREADS * k;
        const bool val = plane_valid(p, P, N, nv);
        ok |= (val ? 1u : 0u) << k;
        v[k] = xp[p < P ? p : 0];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) s += ((ok >> k) & 1u) ? v[k] : 0.f;
    const float m = (float)nv * (float)nv;
    s = plane_sum(s, sm, tid);
    const float mean = m > 0.f ? s / m : 0.f;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) {
        const float d = ((ok >> k) & 1u) ? v[k] - mean : 0.f;
        s2 += d * d;
    }
    s2 = plane_sum(s2, sm, tid);
    const float4 rec = nrm_record(mean, s2, m, (float)nv, gw ? gw[c] : 1.f, eps);
    if (tid == 0) reinterpret_cast<float4 *>(nrm)[gc] = rec;
    const float be = beta ? beta[c] : 0.f;
    float *yp = y + (long long)g * yg + (long long)c * ldy;
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) {
        const int p = tid + PLANE_THREADS * k;
        if (p < P) yp[p] = ((ok >> k) & 1u) ? (v[k] - rec.x) * rec.y + be : 0.f;
    }
}

// dz = ca*dy + cb*(z - mean) + cc with the coefficients of gn_bwd_coef_kernel formed from this plane's own S1 / S2
__global__ __launch_bounds__(PLANE_THREADS) void gn_plane_bwd_kernel(const float *dy, long long dg, long long ldd, const float *z,
                                                                     long long zg, long long ldz, const float *nrm,
                                                                     const int *nvalid, int C, int N, float *dz, long long og,
                                                                     long long ldo, float *s12) {
    __shared__ float sm[4];
    const int gc = blockIdx.x, g = gc / C, c = gc - g * C, tid = threadIdx.x;
    const int nv = nvalid_of(nvalid, g, N), P = N * N;
    const float4 n = reinterpret_cast<const float4 *>(nrm)[gc];
    const float *dp = dy + (long long)g * dg + (long long)c * ldd;
    const float *zp = z + (long long)g * zg + (long long)c * ldz;
    float d[PLANE_EPT], u[PLANE_EPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) {
        const int p = tid + PLANE_THREADS * k;
        const bool val = plane_valid(p, P, N, nv);
        const float dv = dp[p < P ? p : 0], zv = zp[p < P ? p : 0];
        d[k] = val ? dv : 0.f;
        u[k] = val ? zv - n.x : 0.f;
    }
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) {
        s1 += d[k];
        s2 += d[k] * u[k];
    }
    s1 = plane_sum(s1, sm, tid);
    s2 = plane_sum(s2, sm, tid);
    if (tid == 0) {
        s12[(long long)gc * 2] = s1;
        s12[(long long)gc * 2 + 1] = s2;
    }
    const float m = (float)nv * (float)nv;
    const float ca = n.y, cb = m > 0.f ? -n.y * s2 * n.w / m : 0.f, cc = m > 0.f ? -n.y * s1 / m : 0.f;
    float *op = dz + (long long)g * og + (long long)c * ldo;
#pragma unroll
    for (int k = 0; k < PLANE_EPT; ++k) {
        const int p = tid + PLANE_THREADS * k;
        if (p < P) op[p] = plane_valid(p, P, N, nv) ? ca * d[k] + cb * u[k] + cc : 0.f;
    }
}
}  // namespace

extern "C" int fgnn_gn_plane_supported(int N) { return (long long)N * N <= PLANE_MAX ? 1 : 0; }

extern "C" int fgnn_gn_plane_fwd(const float *x, long long gstride, long long ldp, const float *gn_weight, const float *beta,
                                 const int *nvalid, int G, int C, int N, float eps, float *y, long long ygstride, long long ldy,
                                 float *nrm, void *stream) {
    FGNN_CHECK(x && y && nrm && G > 0 && C > 0 && N > 0 && ldp >= (long long)N * N && ldy >= (long long)N * N,
               "fgnn_gn_plane_fwd: bad arguments");
    FGNN_CHECK(fgnn_gn_plane_supported(N), "fgnn_gn_plane_fwd: N*N = %d > %d (use fgnn_gn_stats + fgnn_gn_apply)", N * N, PLANE_MAX);
    hipLaunchKernelGGL(gn_plane_fwd_kernel, dim3(G * C), dim3(PLANE_THREADS), 0, (hipStream_t)stream, x, gstride, ldp, gn_weight,
                       beta, nvalid, C, N, eps, y, ygstride, ldy, nrm);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_gn_plane_bwd(const float *dy, long long dgstride, long long ldd, const float *z, long long zgstride,
                                 long long ldz, const float *nrm, const int *nvalid, int G, int C, int N, float *dz,
                                 long long ogstride, long long ldo, float *s12, float *dgn_w, float *dgn_b, void *stream) {
    FGNN_CHECK(dy && z && nrm && dz && s12 && G > 0 && C > 0 && N > 0, "fgnn_gn_plane_bwd: bad arguments");
    FGNN_CHECK(fgnn_gn_plane_supported(N), "fgnn_gn_plane_bwd: N*N = %d > %d (use fgnn_gn_bwd_stats / _coef / _apply)", N * N,
               PLANE_MAX);
    hipLaunchKernelGGL(gn_plane_bwd_kernel, dim3(G * C), dim3(PLANE_THREADS), 0, (hipStream_t)stream, dy, dgstride, ldd, z, zgstride,
                       ldz, nrm, nvalid, C, N, dz, ogstride, ldo, s12);
    FGNN_LAUNCH_CHECK();
    if (dgn_w || dgn_b) {
        hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s12, nrm, G, C, dgn_w, dgn_b);
        FGNN_LAUNCH_CHECK();
    }
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
    CoefJobs J = {{s12, nullptr}, {nrm, nullptr}, {coef, nullptr}};
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((tot + 255) / 256, 1), dim3(256), 0, (hipStream_t)stream, J,
                       nvalid, G, C, N);
    FGNN_LAUNCH_CHECK();
    if (dgn_w || dgn_b) {
        hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s12, nrm, G, C,
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
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((count + 63) / 64), dim3(256), 0, (hipStream_t)stream, wpart, num_wg,
                       count, 1.0f, out);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_sum_scale(const float *in, int rows, int cols, float scale, float *out, void *stream) {
    FGNN_CHECK(in && out && rows > 0 && cols > 0, "fgnn_sum_scale: bad arguments");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, in, rows, cols,
                       scale, out);
    FGNN_LAUNCH_CHECK();
    return 0;
}

static int gn_bwd_coef_tiles_launch(const float *s12part, const float *nrm, const int *nvalid, int G, int C, int N, int tpg,
                                    float *s12, float *coef, int tr, void *stream) {
    FGNN_CHECK(s12part && nrm && s12 && coef && G > 0 && C > 0 && N > 0 && tpg > 0, "fgnn_gn_bwd_coef_tiles: bad arguments");
    const int tot = G * C;
    hipLaunchKernelGGL(gn_bwd_coef_tiles_kernel, dim3((tot + 3) / 4), dim3(256), 0, (hipStream_t)stream, s12part, nrm,
                       nvalid, G, C, N, tpg, s12, coef, tr);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_gn_bwd_coef_tiles_tpg(const float *s12part, const float *nrm, const int *nvalid, int G, int C, int N,
                                          int tpg, float *s12, float *coef, void *stream) {
    return gn_bwd_coef_tiles_launch(s12part, nrm, nvalid, G, C, N, tpg, s12, coef, 1, stream);
}
extern "C" int fgnn_gn_bwd_coef_tiles(const float *s12part, const float *nrm, const int *nvalid, int G, int C, int N,
                                      float *s12, float *coef, void *stream) {
    return gn_bwd_coef_tiles_launch(s12part, nrm, nvalid, G, C, N, fgnn_tiles_per_graph(N), s12, coef, 0, stream);
}

extern "C" int fgnn_grad_finalize(const fgnn_grad_job *jobs, int njobs, int num_wg, int G, int C, void *stream) {
    FGNN_CHECK(jobs && njobs > 0 && njobs <= FGNN_MAX_GRAD_JOBS && num_wg > 0, "fgnn_grad_finalize: bad arguments (njobs=%d)", njobs);
    GradJobs J;
    int maxc = 0;
    for (int i = 0; i < njobs; ++i) {
        FGNN_CHECK(jobs[i].wpart && jobs[i].out && jobs[i].count > 0, "fgnn_grad_finalize: job %d incomplete", i);
        FGNN_CHECK(!jobs[i].s12 || jobs[i].nrm, "fgnn_grad_finalize: job %d has s12 but no nrm", i);
        J.job[i] = jobs[i];
        if (jobs[i].count > maxc) maxc = jobs[i].count;
    }
    hipLaunchKernelGGL(grad_finalize_kernel, dim3((maxc + 63) / 64 + 1, njobs), dim3(256), 0, (hipStream_t)stream, J,
                       num_wg, G, C);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_gn_bwd_coef2(const float *s12_0, const float *s12_1, const float *nrm0, const float *nrm1,
                                 const int *nvalid, int G, int C, int N, float *coef0, float *coef1, void *stream) {
    FGNN_CHECK(s12_0 && s12_1 && nrm0 && nrm1 && coef0 && coef1 && G > 0 && C > 0 && N > 0, "fgnn_gn_bwd_coef2: bad arguments");
    const int tot = G * C;
    CoefJobs J = {{s12_0, s12_1}, {nrm0, nrm1}, {coef0, coef1}};
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((tot + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, J, nvalid, G, C, N);
    FGNN_LAUNCH_CHECK();
    return 0;
}
