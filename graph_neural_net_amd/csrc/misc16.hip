// Small kernels of the bf16 path (gfx950): operand-image packing, fp32 <-> bf16 slab conversion with the padded row
// pitch, ColumnMaxPooling (models/layers.py:194-203) forward / backward on bf16 slabs.
#include <float.h>
#include "fgnn_bf16.h"

namespace {

// grid (blocks per job, njobs): one thread per (step, lane) writes the lane's 4 dwords; the tail is plain fp32
__global__ __launch_bounds__(256) void pack16_kernel(const Pack16Jobs J) {
    pack16_job_body(J.job[blockIdx.y], blockIdx.x, gridDim.x, threadIdx.x);
}

__global__ __launch_bounds__(256) void to_bf16_kernel(const float *x, const int *nvalid, int C, int N, int ldr,
                                                      unsigned short *y, long long gstride, long long ldp) {
    const int gc = blockIdx.y, g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    const float *src = x + (long long)gc * N * N;
    unsigned short *dst = y + (long long)g * gstride + (long long)c * ldp;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < N * ldr; p += gridDim.x * 256) {
        const int i = p / ldr, j = p - i * ldr;
        const float v = (i < nv && j < nv) ? src[(long long)i * N + j] : 0.f;
        dst[p] = (unsigned short)(cvt_pk(v, 0.f) & 0xffffu);
    }
}

__global__ __launch_bounds__(256) void from_bf16_kernel(const unsigned short *y, long long gstride, long long ldp, int C, int N,
                                                        int ldr, float *x) {
    const int gc = blockIdx.y, g = gc / C, c = gc - g * C;
    const unsigned short *src = y + (long long)g * gstride + (long long)c * ldp;
    float *dst = x + (long long)gc * N * N;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < N * N; p += gridDim.x * 256) {
        const int i = p / N, j = p - i * N;
        dst[p] = __builtin_bit_cast(float, (unsigned)src[(long long)i * ldr + j] << 16);
    }
}

// 16 lanes per (g, c, i) row (4 rows per wave, 16 per workgroup): a lane walks the row in 16-byte pieces of eight
// elements; first maximum (torch.max semantics: ties -> lower index)
__global__ __launch_bounds__(256) void colmax_fwd16_kernel(const fgnn_slab16 y, const int *nvalid, int G, int N, int ldr,
                                                           float *e, int *idx) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const long long t = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int C = y.C;
    const int sub = threadIdx.x & 15;
    const bool live = t < (long long)G * C * N;
    const long long tt = live ? t : 0;
    const int i = (int)(tt % N);
    const int gc = (int)(tt / N);
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    float best = -FLT_MAX;
    int bi = 0x7fffffff;
    if (live && i < nv) {
        const u32x4 *row = reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned short *>(y.ptr) +
                                                            (long long)g * y.gstride + (long long)c * y.ldp + (long long)i * ldr);
        float a = 1.f, b = 0.f;
        if (y.nrm) {
            const float4 n = reinterpret_cast<const float4 *>(y.nrm)[gc];
            a = n.y;
            b = (y.beta ? y.beta[c] : 0.f) - n.x * n.y;
        }
        for (int p = sub; 8 * p < nv; p += 16) {
            const u32x4 d = row[p];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 8 * p + 2 * q;
                const float v0 = fmaf(bf_lo(d[q]), a, b), v1 = fmaf(bf_hi(d[q]), a, b);
                if (j < nv && v0 > best) {
                    best = v0;
                    bi = j;
                }
                if (j + 1 < nv && v1 > best) {
                    best = v1;
                    bi = j + 1;
                }
            }
        }
    }
    for (int o = 8; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (live && sub == 0) {
        const bool ok = i < nv;
        e[t] = ok ? best : 0.f;
        idx[t] = ok ? bi : 0;
    }
}

// one workgroup per (g,c): stage the rows' (argmax, R(de)), then walk the channel linearly in pixel pairs
constexpr int CMB16_MAXN = 1024;
__global__ __launch_bounds__(256) void colmax_bwd16_kernel(const float *de, const int *idx, const int *nvalid, int G, int C,
                                                           int N, int ldr, unsigned short *dy, long long gstride,
                                                           long long ldp, const fgnn_slab16 y, float *s12, float *coef) {
    __shared__ int sidx[CMB16_MAXN];
    __shared__ unsigned sde[CMB16_MAXN];
    __shared__ float red[4][2];
    const int gc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = gc / C, c = gc - g * C;
    const int nv = nvalid_of(nvalid, g, N);
    unsigned *mat = reinterpret_cast<unsigned *>(dy + (long long)g * gstride + (long long)c * ldp);
    const unsigned short *zm =
        s12 ? reinterpret_cast<const unsigned short *>(y.ptr) + (long long)g * y.gstride + (long long)c * y.ldp : nullptr;
    const float mean = s12 ? y.nrm[(long long)gc * 4] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    // N <= 256 (one row per thread): the z value at the arg-max position -- a second, dependent memory round trip -- is only
    // requested here and consumed after the store sweep
    const bool late = N <= 256;
    unsigned short zlate = 0;
    float dlate = 0.f;
    for (int i = tid; i < N; i += 256) {
        const long long t = (long long)gc * N + i;
        const int bi = idx[t];
        const unsigned db = i < nv ? (cvt_pk(de[t], 0.f) & 0xffffu) : 0u;        // R(de)
        sidx[i] = bi;
        sde[i] = db;
        if (s12 && i < nv) {
            const float d = bf_lo(db);
            if (late) {
                zlate = zm[(long long)i * ldr + bi];
                dlate = d;
            } else {
                s1 += d;
                s2 += d * (bf_lo(zm[(long long)i * ldr + bi]) - mean);
            }
        }
    }
    __syncthreads();
    // the channel is written in 16-byte pieces of eight pixels (ldr is a multiple of 8, the channel base of 64 elements)
    const int hp4 = ldr / 8;
    uint4 *mat4 = reinterpret_cast<uint4 *>(mat);
    for (int q = tid; q < N * hp4; q += 256) {
        const int i = q / hp4, p8 = q - i * hp4;
        const int rel = sidx[i] - 8 * p8;              // 0..7: the arg-max column lies in this piece
        const unsigned d = sde[i];
        const unsigned w = (rel & 1) ? (d << 16) : d;
        const bool in = (unsigned)rel < 8u;
        const int k = rel >> 1;
        uint4 v;
        v.x = (in && k == 0) ? w : 0u;
        v.y = (in && k == 1) ? w : 0u;
        v.z = (in && k == 2) ? w : 0u;
        v.w = (in && k == 3) ? w : 0u;
        mat4[q] = v;
    }
    if (s12) {
        if (late && tid < nv) {
            s1 += dlate;
            s2 += dlate * (bf_lo(zlate) - mean);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            red[wave][0] = s1;
            red[wave][1] = s2;
        }
        __syncthreads();
        if (tid < 2) {
            const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            s12[(long long)gc * 2 + tid] = v;
            // dz coefficients of the producing MLP (the arithmetic of gn_bwd_coef_kernel), saving that launch
            const float v1 = __shfl(v, 0), v2 = __shfl(v, 1);
            if (coef && tid == 0) {
                const float4 n = reinterpret_cast<const float4 *>(y.nrm)[gc];
                const float m = (float)nv * (float)nv;
                float4 o;
                o.x = n.x;
                o.y = n.y;
                o.z = m > 0.f ? -n.y * v2 * n.w / m : 0.f;
                o.w = m > 0.f ? -n.y * v1 / m : 0.f;
                reinterpret_cast<float4 *>(coef)[gc] = o;
            }
        }
    }
}

}  // namespace

extern "C" int fgnn_pack16_floats(int kind, int ca, int cb, int depth, int nmlp) {
    return pk16_layout(kind, ca, cb, depth).floats * (kind == 0 ? nmlp : 1);
}

extern "C" int fgnn_pack16_operands(const fgnn_pack_job *jobs, int njobs, void *stream) {
    FGNN_CHECK(jobs && njobs > 0 && njobs <= FGNN_MAX_PACK_JOBS, "fgnn_pack16_operands: bad arguments (njobs=%d)", njobs);
    Pack16Jobs J;
    for (int i = 0; i < njobs; ++i) {
        FGNN_CHECK(jobs[i].out && jobs[i].depth >= 2 && jobs[i].depth <= FGNN_MAX_DEPTH && (jobs[i].nmlp == 1 || jobs[i].nmlp == 2),
                   "fgnn_pack16_operands: job %d malformed", i);
        FGNN_CHECK((jobs[i].ca == 2 || jobs[i].ca == 32) && (jobs[i].cb == 0 || jobs[i].cb == 2 || jobs[i].cb == 32),
                   "fgnn_pack16_operands: job %d: slab widths must be 2 or 32", i);
        J.job[i] = jobs[i];
    }
    hipLaunchKernelGGL(pack16_kernel, dim3(PACK16_BLOCKS_PER_JOB, njobs), dim3(256), 0, (hipStream_t)stream, J);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_to_bf16(const float *x, const int *nvalid, int G, int C, int N, int ldr, void *y, long long gstride,
                            long long ldp, void *stream) {
    FGNN_CHECK(x && y && G > 0 && C > 0 && N > 0 && ldr >= N && ldp >= (long long)N * ldr, "fgnn_to_bf16: bad arguments");
    const int nb = (N * ldr + 255) / 256;
    hipLaunchKernelGGL(to_bf16_kernel, dim3(nb > 64 ? 64 : nb, G * C), dim3(256), 0, (hipStream_t)stream, x, nvalid, C, N, ldr,
                       (unsigned short *)y, gstride, ldp);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_from_bf16(const void *y, long long gstride, long long ldp, int G, int C, int N, int ldr, float *x,
                              void *stream) {
    FGNN_CHECK(x && y && G > 0 && C > 0 && N > 0 && ldr >= N, "fgnn_from_bf16: bad arguments");
    const int nb = (N * N + 255) / 256;
    hipLaunchKernelGGL(from_bf16_kernel, dim3(nb > 64 ? 64 : nb, G * C), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)y, gstride, ldp, C, N, ldr, x);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_colmax_fwd16(const fgnn_slab16 *y, const int *nvalid, int G, int N, int ldr, float *e, int *idx,
                                 void *stream) {
    FGNN_CHECK(y && y->ptr && e && idx && G > 0 && N > 0 && ldr >= N && ldr % 8 == 0 && y->ldp % 8 == 0 && y->gstride % 8 == 0,
               "fgnn_colmax_fwd16: bad arguments (row pitch / strides must be multiples of 8 elements)");
    const long long rows = (long long)G * y->C * N;
    hipLaunchKernelGGL(colmax_fwd16_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *y, nvalid, G, N,
                       ldr, e, idx);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_colmax_bwd16_coef(const float *de, const int *idx, const int *nvalid, int G, int C, int N, int ldr, void *dy,
                                      long long gstride, long long ldp, const fgnn_slab16 *y, float *s12, float *coef,
                                      void *stream) {
    FGNN_CHECK(de && idx && dy && G > 0 && C > 0 && N > 0 && ldr >= N && ldr % 2 == 0, "fgnn_colmax_bwd16: bad arguments");
    FGNN_CHECK(N <= CMB16_MAXN, "fgnn_colmax_bwd16: N=%d > %d", N, CMB16_MAXN);
    FGNN_CHECK(!s12 || (y && y->ptr && y->nrm), "fgnn_colmax_bwd16: s12 needs the normalised input slab");
    FGNN_CHECK(!coef || s12, "fgnn_colmax_bwd16_coef: the coefficients come with the s12 sums");
    fgnn_slab16 none = {};
    hipLaunchKernelGGL(colmax_bwd16_kernel, dim3(G * C), dim3(256), 0, (hipStream_t)stream, de, idx, nvalid, G, C, N, ldr,
                       (unsigned short *)dy, gstride, ldp, y ? *y : none, s12, coef);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_colmax_bwd16(const float *de, const int *idx, const int *nvalid, int G, int C, int N, int ldr, void *dy,
                                 long long gstride, long long ldp, const fgnn_slab16 *y, float *s12, void *stream) {
    return fgnn_colmax_bwd16_coef(de, idx, nvalid, G, C, N, ldr, dy, gstride, ldp, y, s12, nullptr, stream);
}
