// Generic-width 1x1 convolution (any Cin, any Cout) for gfx950: the building block of MlpBlock_Real for channel widths the
// fused 32-wide kernels (mlp_fwd.hip / mlp_bwd.hip) are not built for.  Replaces, per layer,
//   out = activation(conv_layer(out))       models/layers.py:125-131 (nn.Conv2d(k=1, bias=True) + F.relu)
// and its autograd backward.  fp32 storage, v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain over the input channels).
//
//   conv1x1_kernel      y[g][o][p] = act( b[o] + sum_k W[o][k] * x[g][k][p] )         forward, and (with W^T given by
//                       strides, an optional ReLU mask on the input, no bias) the input gradient
//   conv1x1_dw_kernel   dW[o][c] = sum_{g,p} dz[g][o][p] * x[g][c][p],  db[o] = sum_{g,p} dz[g][o][p]
//                       per-chunk partials, finished by fgnn_reduce_partials in a fixed order (bit-reproducible)
//
// A tile is 32 consecutive pixels of one graph times 32 output channels.  Pixels outside the valid n x n corner of a
// ragged graph produce exact zeros (the MaskedTensor re-mask of maskedtensors/maskedtensor.py:98-112) and contribute
// nothing to the parameter gradients.
#include "fgnn_common.h"

namespace {

constexpr int CONV_WAVES = 4;
constexpr int CONV_T = 32;          // pixels per tile
constexpr int CONV_TLD = 33;        // LDS row stride of the transposing tiles (floats)

struct TileCtx {
    int g, p;
    bool inb, ok;
};
// tile t -> graph, this lane's pixel, in-bounds (p < N*N) and valid (inside the graph's n x n corner)
DEVI TileCtx tile_ctx(int t, int tpg, int N, const int *nvalid, int col) {
    TileCtx c;
    c.g = __builtin_amdgcn_readfirstlane(t / tpg);
    c.p = (t - c.g * tpg) * CONV_T + col;
    const int nv = nvalid_of(nvalid, c.g, N);
    const int i = c.p / N, j = c.p - i * N;
    c.inb = c.p < N * N;
    c.ok = c.inb && i < nv && j < nv;
    return c;
}

__global__ __launch_bounds__(64 * CONV_WAVES) void conv1x1_kernel(
    const float *xp, long long x_gs, long long x_ld, const float *mp, const float *W, long long w_so, long long w_sk,
    const float *bias, int relu, const int *nvalid, int G, int N, int M, int K, float *yp, long long y_gs, long long y_ld,
    int tpg, int ntiles) {
    extern __shared__ float wl[];          // this output group's weights, [k][32 outputs], k padded to even
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    const int K2 = (K + 1) / 2, MG = (M + 31) / 32;
    const View xv = make_view(xp, x_gs, x_ld, G), yv = make_view(yp, y_gs, y_ld, G);
    const View mv = mp ? make_view(mp, x_gs, x_ld, G) : xv;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);

    for (int og = 0; og < MG; ++og) {
        __syncthreads();
        for (int e = tid; e < K2 * 64; e += 64 * CONV_WAVES) {
            const int k = e >> 5, o = og * 32 + (e & 31);
            wl[e] = (k < K && o < M) ? W[o * w_so + k * w_sk] : 0.f;
        }
        __syncthreads();
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = og * 32 + ch_of(r, h);
            bv[r] = (bias && o < M) ? bias[o] : 0.f;
        }
        for (int t = t_begin + wv; t < t_end; t += CONV_WAVES) {
            const TileCtx c = tile_ctx(t, tpg, N, nvalid, col);
            const int sx = c.g * xv.gs4, sy = c.g * yv.gs4;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bv[r];
            for (int k0 = 0; k0 < K2; k0 += 4) {
                float b[4], m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 2 * (k0 + u) + h;
                    const int off = (c.inb && k < K) ? k * xv.ld4 + c.p * 4 : OOB_OFF;
                    b[u] = buf_load(xv, off, sx);
                    m[u] = mp ? buf_load(mv, off, sx) : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u < K2) {          // uniform; the padded k of an odd K reads a zero weight and a zero input
                        const float a = wl[(2 * (k0 + u) + h) * 32 + col];
                        acc = mfma32(a, m[u] > 0.f ? b[u] : 0.f, acc);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = og * 32 + ch_of(r, h);
                float v = relu ? fmaxf(acc[r], 0.f) : acc[r];
                v = c.ok ? v : 0.f;
                buf_store(v, yv, (c.inb && o < M) ? o * yv.ld4 + c.p * 4 : OOB_OFF, sy);
            }
        }
    }
}

// grid (chunks, MG*KG): workgroup (chunk, pair) accumulates the 32 x 32 block (og, cg) of dW over its chunk of tiles.
__global__ __launch_bounds__(64 * CONV_WAVES) void conv1x1_dw_kernel(
    const float *dyp, long long d_gs, long long d_ld, const float *mp, const float *xp, long long x_gs, long long x_ld,
    const int *nvalid, int G, int N, int M, int K, float *wpart, int tpg, int ntiles) {
    __shared__ float ta[CONV_WAVES][32 * CONV_TLD], tb[CONV_WAVES][32 * CONV_TLD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    const int KG = (K + 31) / 32;
    const int og = blockIdx.y / KG, cg = blockIdx.y - og * KG;
    const View dv = make_view(dyp, d_gs, d_ld, G), xv = make_view(xp, x_gs, x_ld, G);
    const View mv = mp ? make_view(mp, d_gs, d_ld, G) : dv;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    for (int t = t_begin + wv; t < t_end; t += CONV_WAVES) {
        const TileCtx c = tile_ctx(t, tpg, N, nvalid, col);
        const int sd = c.g * dv.gs4, sx = c.g * xv.gs4;
        // coalesced loads ([channel row][32 pixels]), transposed through this wave's LDS tiles
        float d[16], m[16], x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = og * 32 + 2 * r + h, cc = cg * 32 + 2 * r + h;
            const int od = (c.ok && o < M) ? o * dv.ld4 + c.p * 4 : OOB_OFF;
            d[r] = buf_load(dv, od, sd);
            m[r] = mp ? buf_load(mv, od, sd) : 1.f;
            x[r] = buf_load(xv, (c.ok && cc < K) ? cc * xv.ld4 + c.p * 4 : OOB_OFF, sx);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ta[wv][(2 * r + h) * CONV_TLD + col] = m[r] > 0.f ? d[r] : 0.f;
            tb[wv][(2 * r + h) * CONV_TLD + col] = x[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = ta[wv][col * CONV_TLD + 2 * s + h];       // dz[o = col][pixel 2s + h]
            const float b = tb[wv][col * CONV_TLD + 2 * s + h];       // x [c = col][pixel 2s + h]
            acc = mfma32(a, b, acc);
            bsum += a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the four waves' blocks summed in wave order
    __syncthreads();
    float *red = &ta[0][0];          // [wave][16 * 64]   (4 * 1056 floats available)
    float *redb = &tb[0][0];         // [wave][64]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wv * 1024 + r * 64 + lane] = acc[r];
    redb[wv * 64 + lane] = bsum;
    __syncthreads();
    const long long cnt = (long long)M * K + M;
    float *out = wpart + blockIdx.x * cnt;
    for (int e = tid; e < 1024; e += 64 * CONV_WAVES) {
        const int r = e >> 6, l = e & 63;
        const float s = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        const int o = og * 32 + ch_of(r, l >> 5), cc = cg * 32 + (l & 31);
        if (o < M && cc < K) out[(long long)o * K + cc] = s;
    }
    if (cg == 0 && tid < 32) {
        float s = 0.f;
        for (int w = 0; w < CONV_WAVES; ++w) s += redb[w * 64 + tid] + redb[w * 64 + 32 + tid];
        const int o = og * 32 + tid;
        if (o < M) out[(long long)M * K + o] = s;
    }
}

int conv_tiles(int G, int N, int *tpg) {
    *tpg = (N * N + CONV_T - 1) / CONV_T;
    return G * *tpg;
}

}  // namespace

extern "C" int fgnn_conv1x1(const float *x, long long x_gstride, long long x_ld, const float *relu_mask, const float *W,
                            long long w_ostride, long long w_kstride, const float *bias, int relu, const int *nvalid, int G,
                            int N, int M, int K, float *y, long long y_gstride, long long y_ld, void *stream) {
    FGNN_CHECK(x && W && y && G > 0 && N > 0 && M > 0 && K > 0, "fgnn_conv1x1: bad arguments");
    FGNN_CHECK(K <= FGNN_CONV_MAX_CH && M <= FGNN_CONV_MAX_CH, "fgnn_conv1x1: at most %d channels (got %d -> %d)", FGNN_CONV_MAX_CH, K, M);
    const long long P = (long long)N * N, lim = (1ll << 31) / 4;
    FGNN_CHECK(x_ld >= P && y_ld >= P && x_gstride >= K * x_ld && y_gstride >= M * y_ld, "fgnn_conv1x1: strides smaller than the tensors");
    FGNN_CHECK(G * x_gstride < lim && G * y_gstride < lim, "fgnn_conv1x1: tensors of 2 GiB or more are not supported");
    int tpg;
    const long long nt = (long long)G * ((P + CONV_T - 1) / CONV_T);
    FGNN_CHECK(nt < (1ll << 30), "fgnn_conv1x1: too many tiles");
    const int ntiles = conv_tiles(G, N, &tpg);
    int grid = (ntiles + 2 * CONV_WAVES - 1) / (2 * CONV_WAVES);
    if (grid > 1024) grid = 1024;
    const size_t lds = (size_t)((K + 1) / 2) * 64 * sizeof(float);
    hipLaunchKernelGGL(conv1x1_kernel, dim3(grid), dim3(64 * CONV_WAVES), lds, (hipStream_t)stream, x, x_gstride, x_ld,
                       relu_mask, W, w_ostride, w_kstride, bias, relu, nvalid, G, N, M, K, y, y_gstride, y_ld, tpg, ntiles);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_conv1x1_dw_chunks(int G, int N) {
    int tpg;
    const int ntiles = conv_tiles(G, N, &tpg);
    int chunks = (ntiles + 4 * CONV_WAVES - 1) / (4 * CONV_WAVES);
    return chunks > 128 ? 128 : (chunks < 1 ? 1 : chunks);
}

extern "C" int fgnn_conv1x1_dw(const float *dy, long long d_gstride, long long d_ld, const float *relu_mask, const float *x,
                               long long x_gstride, long long x_ld, const int *nvalid, int G, int N, int M, int K, float *wpart,
                               void *stream) {
    FGNN_CHECK(dy && x && wpart && G > 0 && N > 0 && M > 0 && K > 0, "fgnn_conv1x1_dw: bad arguments");
    FGNN_CHECK(K <= FGNN_CONV_MAX_CH && M <= FGNN_CONV_MAX_CH, "fgnn_conv1x1_dw: at most %d channels (got %d -> %d)", FGNN_CONV_MAX_CH, K, M);
    const long long P = (long long)N * N, lim = (1ll << 31) / 4;
    FGNN_CHECK(x_ld >= P && d_ld >= P && x_gstride >= K * x_ld && d_gstride >= M * d_ld, "fgnn_conv1x1_dw: strides smaller than the tensors");
    FGNN_CHECK(G * x_gstride < lim && G * d_gstride < lim, "fgnn_conv1x1_dw: tensors of 2 GiB or more are not supported");
    FGNN_CHECK((long long)G * ((P + CONV_T - 1) / CONV_T) < (1ll << 30), "fgnn_conv1x1_dw: too many tiles");
    int tpg;
    const int ntiles = conv_tiles(G, N, &tpg);
    const int chunks = fgnn_conv1x1_dw_chunks(G, N);
    const int pairs = ((M + 31) / 32) * ((K + 31) / 32);
    hipLaunchKernelGGL(conv1x1_dw_kernel, dim3(chunks, pairs), dim3(64 * CONV_WAVES), 0, (hipStream_t)stream, dy, d_gstride,
                       d_ld, relu_mask, x, x_gstride, x_ld, nvalid, G, N, M, K, wpart, tpg, ntiles);
    FGNN_LAUNCH_CHECK();
    return 0;
}
