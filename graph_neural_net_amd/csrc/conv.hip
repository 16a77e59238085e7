// Generic-width 1x1 convolution (any Cin, any Cout) for gfx950: the building block of MlpBlock_Real for channel widths the
// fused 32-wide kernels (mlp_fwd.hip / mlp_bwd.hip) are not built for.  Replaces, per layer,
//   out = activation(conv_layer(out))       models/layers.py:125-131 (nn.Conv2d(k=1, bias=True) + F.relu)
// and its autograd backward.  fp32 storage, v_mfma_f32_32x32x2_f32 (an exact fp32 fma chain over the input channels).
//
//   conv1x1_pre_kernel  y[g][o][p] = act( b[o] + sum_k W[o][k] * x[g][k][p] )         forward, and (with W^T given by
//                       strides, an optional ReLU mask on the input, no bias) the input gradient.  K <= 128: a wave requests
//                       ALL input channels of its tile at once (K/2 loads in flight per lane), then runs every 32-wide
//                       output group from those registers -- the input is read once, whatever M is
//   conv1x1_kernel      the same for 128 < K <= 256: input streamed in groups of 8 channels, once per output group
//   conv1x1_dw_kernel   dW[o][c] = sum_{g,p} dz[g][o][p] * x[g][c][p],  db[o] = sum_{g,p} dz[g][o][p]
//                       per-chunk partials, finished by fgnn_reduce_partials in a fixed order (bit-reproducible).  A workgroup
//                       owns a 2 x 2 set of 32 x 32 blocks of dW: up to 64 x 64 channels dz and x are read once
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

// K <= 2 * KH input channels held in registers.  LDS: weights [og][k][32 outputs] (k padded to even), then the biases.
template <int KH>
struct ConvTile {
    float b[KH];
    TileCtx c;
};
template <int KH>
DEVI void conv_request(ConvTile<KH> &T, int t, const View &xv, const View &mv, bool masked, int tpg, int N, const int *nvalid,
                       int K, int K2, int col, int h) {
    T.c = tile_ctx(t, tpg, N, nvalid, col);
    const int sx = T.c.g * xv.gs4;
    // lane (pixel col, half h) reads channel k = 2u + h: the pixel / half part of the address in ONE VGPR, the 2u rows in
    // the scalar offset; only the last pair of an odd K needs its own (half 1 is past K)
    const int voff = T.c.inb ? h * xv.ld4 + T.c.p * 4 : OOB_OFF;
    const int voff_last = (T.c.inb && 2 * (K2 - 1) + h < K) ? voff : OOB_OFF;
#pragma unroll
    for (int u = 0; u < KH; ++u) {
        const int vo = u < K2 - 1 ? voff : (u == K2 - 1 ? voff_last : OOB_OFF);     // uniform selects
        T.b[u] = buf_load(xv, vo, sx + 2 * u * xv.ld4);
        if (masked) {
            const float m = buf_load(mv, vo, sx + 2 * u * xv.ld4);
            T.b[u] = m > 0.f ? T.b[u] : 0.f;
        }
    }
}
template <int KH>
DEVI void conv_multiply(ConvTile<KH> &T, const float *wl, const float *bl, const View &yv, int relu, int M, int K2,
                        int MG, int col, int h) {
    const int sy = T.c.g * yv.gs4;
    for (int og = 0; og < MG; ++og) {
        const float *wg = wl + og * K2 * 64 + h * 32 + col;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bl[og * 32 + ch_of(r, h)];
#pragma unroll
        for (int u = 0; u < KH; ++u) {
            if (u < K2) acc = mfma32(wg[u * 64], T.b[u], acc);     // uniform; same k order as the streaming kernel
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = og * 32 + ch_of(r, h);
            float v = relu ? fmaxf(acc[r], 0.f) : acc[r];
            v = T.c.ok ? v : 0.f;
            buf_store(v, yv, (T.c.inb && o < M) ? o * yv.ld4 + T.c.p * 4 : OOB_OFF, sy);
        }
    }
}
template <int KH>
__global__ __launch_bounds__(64 * CONV_WAVES) void conv1x1_pre_kernel(
    const float *xp, long long x_gs, long long x_ld, const float *mp, const float *W, long long w_so, long long w_sk,
    const float *bias, int relu, const int *nvalid, int G, int N, int M, int K, float *yp, long long y_gs, long long y_ld,
    int tpg, int ntiles) {
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    const int K2 = (K + 1) / 2, MG = (M + 31) / 32;
    float *bl = wl + MG * K2 * 64;
    const View xv = make_view(xp, x_gs, x_ld, G), yv = make_view(yp, y_gs, y_ld, G);
    const bool masked = mp != nullptr;
    const View mv = masked ? make_view(mp, x_gs, x_ld, G) : xv;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    ConvTile<KH> A;
    int t = t_begin + wv;
    if (t < t_end) conv_request<KH>(A, t, xv, mv, masked, tpg, N, nvalid, K, K2, col, h);     // in flight under the weight copy
    for (int e = tid; e < MG * K2 * 64; e += 64 * CONV_WAVES) {
        const int og = e / (K2 * 64), rem = e - og * (K2 * 64);
        const int k = rem >> 5, o = og * 32 + (rem & 31);
        wl[e] = (k < K && o < M) ? W[o * w_so + k * w_sk] : 0.f;
    }
    for (int e = tid; e < MG * 32; e += 64 * CONV_WAVES) bl[e] = (bias && e < M) ? bias[e] : 0.f;
    __syncthreads();
    // (requesting tile t + 1 ahead of tile t's products -- two register sets -- measured slower: 47.6 against 40.3 us at
    //  64 -> 64 channels, B = 32, N = 50; the occupancy the second set costs is worth more than the overlap it buys)
    for (; t < t_end; t += CONV_WAVES) {
        if (t != t_begin + wv) conv_request<KH>(A, t, xv, mv, masked, tpg, N, nvalid, K, K2, col, h);
        conv_multiply<KH>(A, wl, bl, yv, relu, M, K2, MG, col, h);
    }
}

// ---- chain of up to three 1x1 convolutions, intermediates in registers (fgnn_conv_chain) ------------------------------------
// Layer 0 reads its input like conv1x1_pre_kernel (lane = pixel, half h, channel 2u + h).  A layer's output is a set of
// 32-channel D fragments (lane = pixel, register r = channel ch_of(r, h) of the group); used as the next layer's B operand,
// k-step (ig, r) contracts channels 32 ig + ch_of(r, 0 / 1) -- the operand image of that layer is laid out accordingly, so the
// activations never move.  LDS: [image 0 | image 1 | image 2 | biases], image 0 = [og][k][32 o], image l >= 1 =
// [og][ig][r][h][32 o].  MAXG = 32-channel groups a layer's output may have (2 or 4).
struct ChainDev {
    const float *W[3];
    long long w_so[3], w_sk[3];
    const float *bias[3];
    int M[3], K[3], relu[3];
    const float *mask[3];
    float *out[3];
    long long o_gs[3], o_ld;
};

template <int MAXG>
struct Frags {
    f32x16 f[MAXG];
};

// one layer l >= 1: in = prev fragments (IG groups), out -> cur fragments
template <int MAXG, bool FULL>
DEVI void chain_hidden(Frags<MAXG> &cur, const Frags<MAXG> &prev, const float *wl, const float *bl, int OG, int IG, int col, int h) {
#pragma unroll
    for (int og = 0; og < MAXG; ++og) {
        if (FULL || og < OG) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bl[og * 32 + ch_of(r, h)];
#pragma unroll
            for (int ig = 0; ig < MAXG; ++ig) {
                if (FULL || ig < IG) {
                    const float *wp = wl + ((og * (FULL ? MAXG : IG) + ig) * 16) * 64 + h * 32 + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc = mfma32(wp[r * 64], prev.f[ig][r], acc);
                }
            }
            cur.f[og] = acc;
            __builtin_amdgcn_sched_barrier(0);                  // (keeps the next group's operand reads from being hoisted)
        }
    }
}
// activation, mask, validity, store of one layer's fragments
template <int MAXG, bool FULL>
DEVI void chain_post(Frags<MAXG> &cur, int OG, int M, int relu, const float *mask, float *out, long long o_gs, long long o_ld, int G,
                     const TileCtx &c, int h) {
    const View ov = make_view(out ? out : mask, o_gs, o_ld, G);         // (only dereferenced when the pointer exists)
    const View mv = make_view(mask ? mask : out, o_gs, o_ld, G);
    // the graph / half / pixel part of the address in the VGPR offset, the channel rows (shared by all layers: one channel
    // stride) in the scalar offset
    const int vo = c.inb ? c.g * ov.gs4 + 4 * h * ov.ld4 + c.p * 4 : OOB_OFF;
#pragma unroll
    for (int og = 0; og < MAXG; ++og) {
        if (FULL || og < OG) {
            // channel bound of the group: a full group (every width that is a multiple of 32) needs no per-row predicate
            const bool full = FULL || M - og * 32 >= 32;
            float m[16];
            if (mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ob = og * 32 + (r & 3) + 8 * (r >> 2);
                    m[r] = buf_load(mv, (full || ob + 4 * h < M) ? vo : OOB_OFF, ob * ov.ld4);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = cur.f[og][r];
                if (relu) v = relu1(v);
                if (mask) v = m[r] > 0.f ? v : 0.f;
                v = c.ok ? v : 0.f;
                cur.f[og][r] = v;
                if (out) {
                    const int ob = og * 32 + (r & 3) + 8 * (r >> 2);
                    buf_store(v, ov, (full || ob + 4 * h < M) ? vo : OOB_OFF, ob * ov.ld4);
                }
            }
        }
    }
}

// FULL: K_0 = 2 KH and every layer has exactly MAXG full output groups (the 64-wide model): no bounds inside the tile loop
template <int KH, int MAXG, bool FULL>
__global__ __launch_bounds__(64 * CONV_WAVES, MAXG == 2 ? 2 : 1) void conv_chain_kernel(const float *xp, long long x_gs, long long x_ld, int depth,
                                                                     const ChainDev L, const int *nvalid, int G, int N, int tpg,
                                                                     int ntiles) {
    extern __shared__ float wl[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    const int K2 = (L.K[0] + 1) / 2;
    int OGs[3], off[3];
    int tot = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        OGs[l] = l < depth ? (L.M[l] + 31) / 32 : 0;
        off[l] = tot;
        if (l < depth) tot += l == 0 ? OGs[0] * K2 * 64 : OGs[l] * OGs[l - 1] * 1024;
    }
    float *bl = wl + tot;                       // [layer][128]
    const View xv = make_view(xp, x_gs, x_ld, G);
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    ConvTile<KH> A;
    int t = t_begin + wv;
    if (t < t_end) conv_request<KH>(A, t, xv, xv, false, tpg, N, nvalid, L.K[0], K2, col, h);      // in flight under the image copy
    // operand images
    for (int e = tid; e < OGs[0] * K2 * 64; e += 64 * CONV_WAVES) {
        const int og = e / (K2 * 64), rem = e - og * (K2 * 64);
        const int k = rem >> 5, o = og * 32 + (rem & 31);
        wl[e] = (k < L.K[0] && o < L.M[0]) ? L.W[0][o * L.w_so[0] + k * L.w_sk[0]] : 0.f;
    }
#pragma unroll
    for (int l = 1; l < 3; ++l) {
        if (l < depth) {
            const int IG = OGs[l - 1];
            for (int e = tid; e < OGs[l] * IG * 1024; e += 64 * CONV_WAVES) {
                const int o = e & 31, hh = (e >> 5) & 1, r = (e >> 6) & 15, gi = e >> 10;       // gi = og * IG + ig
                const int og = gi / IG, ig = gi - og * IG;
                const int oc = og * 32 + o, kc = ig * 32 + ch_of(r, hh);
                wl[off[l] + e] = (oc < L.M[l] && kc < L.K[l]) ? L.W[l][oc * L.w_so[l] + kc * L.w_sk[l]] : 0.f;
            }
        }
    }
    for (int e = tid; e < 3 * 128; e += 64 * CONV_WAVES) {
        const int l = e >> 7, o = e & 127;
        bl[e] = (l < depth && L.bias[l] && o < L.M[l]) ? L.bias[l][o] : 0.f;
    }
    __syncthreads();
    for (; t < t_end; t += CONV_WAVES) {
        if (t != t_begin + wv) conv_request<KH>(A, t, xv, xv, false, tpg, N, nvalid, L.K[0], K2, col, h);
        Frags<MAXG> fa, fb;
        // layer 0: from the channel-pair registers
#pragma unroll
        for (int og = 0; og < MAXG; ++og) {
            if (FULL || og < OGs[0]) {
                const float *wg = wl + og * (FULL ? KH : K2) * 64 + h * 32 + col;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bl[og * 32 + ch_of(r, h)];
#pragma unroll
                for (int u = 0; u < KH; ++u) {
                    if (FULL || u < K2) acc = mfma32(wg[u * 64], A.b[u], acc);
                }
                fa.f[og] = acc;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        chain_post<MAXG, FULL>(fa, OGs[0], L.M[0], L.relu[0], L.mask[0], L.out[0], L.o_gs[0], L.o_ld, G, A.c, h);
        if (depth > 1) {
            chain_hidden<MAXG, FULL>(fb, fa, wl + off[1], bl + 128, OGs[1], OGs[0], col, h);
            chain_post<MAXG, FULL>(fb, OGs[1], L.M[1], L.relu[1], L.mask[1], L.out[1], L.o_gs[1], L.o_ld, G, A.c, h);
        }
        if (depth > 2) {
            chain_hidden<MAXG, FULL>(fa, fb, wl + off[2], bl + 256, OGs[2], OGs[1], col, h);
            chain_post<MAXG, FULL>(fa, OGs[2], L.M[2], L.relu[2], L.mask[2], L.out[2], L.o_gs[2], L.o_ld, G, A.c, h);
        }
    }
}

// grid (chunks, sets): workgroup (chunk, set) accumulates a 2 x 2 set of 32 x 32 blocks of dW -- output groups og0, og0 + 1
// times input groups cg0, cg0 + 1 -- over its chunk of tiles; groups past M / K load nothing and their blocks are not stored.
constexpr int DW_TILES = 4;         // LDS transposing tiles per wave: dz of two output groups, x of two input groups
constexpr int DW_WAVES = 8;         // 135 KB of LDS: one workgroup per CU, two waves per SIMD
// `set` = which 2 x 2 set of blocks, `out` = this chunk's partial record [dW (M, K) | db (M)]
DEVI void conv1x1_dw_body(const float *dyp, long long d_gs, long long d_ld, const float *mp, const float *xp, long long x_gs,
                          long long x_ld, const int *nvalid, int G, int N, int M, int K, float *out, int set, int tpg, int ntiles) {
    extern __shared__ float dw_lds[];       // [wave][4 tiles][32 * CONV_TLD]; reused for the final reduction
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    const int KS = ((K + 31) / 32 + 1) / 2;
    const int os = set / KS, cs = set - os * KS;
    const int o0 = os * 64, c0 = cs * 64;                       // first output / input channel of the set
    float *tl = dw_lds + wv * DW_TILES * 32 * CONV_TLD;
    const View dv = make_view(dyp, d_gs, d_ld, G), xv = make_view(xp, x_gs, x_ld, G);
    const View mv = mp ? make_view(mp, d_gs, d_ld, G) : dv;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);

    f32x16 acc[4];                                              // blocks (og, cg) = (0,0) (0,1) (1,0) (1,1)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    for (int t = t_begin + wv; t < t_end; t += DW_WAVES) {
        const TileCtx c = tile_ctx(t, tpg, N, nvalid, col);
        const int sd = c.g * dv.gs4, sx = c.g * xv.gs4;
        // coalesced loads ([channel row][32 pixels]), everything in flight before the first wait
        // (row 2r + h of a group: the half's row and the pixel in the VGPR offset, the 2r rows in the scalar offset)
        float d[2][16], x[2][16];
        const int vd = c.ok ? h * dv.ld4 + c.p * 4 : OOB_OFF, vx = c.ok ? h * xv.ld4 + c.p * 4 : OOB_OFF;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ob = o0 + 32 * q + 2 * r, cb = c0 + 32 * q + 2 * r;       // uniform
                const int od = ob + h < M ? vd : OOB_OFF;
                d[q][r] = buf_load(dv, od, sd + ob * dv.ld4);
                if (mp) {
                    const float m = buf_load(mv, od, sd + ob * dv.ld4);
                    d[q][r] = m > 0.f ? d[q][r] : 0.f;
                }
                x[q][r] = buf_load(xv, cb + h < K ? vx : OOB_OFF, sx + cb * xv.ld4);
            }
        }
        // transposed through this wave's LDS tiles: tile q = dz of output group q, tile 2 + q = x of input group q
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tl[q * 32 * CONV_TLD + (2 * r + h) * CONV_TLD + col] = d[q][r];
                tl[(2 + q) * 32 * CONV_TLD + (2 * r + h) * CONV_TLD + col] = x[q][r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane (channel col, half h), k-step s = pixel 2s + h
        float a[2][16], b[2][16];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                a[q][s] = tl[q * 32 * CONV_TLD + col * CONV_TLD + 2 * s + h];
                b[q][s] = tl[(2 + q) * 32 * CONV_TLD + col * CONV_TLD + 2 * s + h];
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc[0] = mfma32(a[0][s], b[0][s], acc[0]);
            acc[1] = mfma32(a[0][s], b[1][s], acc[1]);
            acc[2] = mfma32(a[1][s], b[0][s], acc[2]);
            acc[3] = mfma32(a[1][s], b[1][s], acc[3]);
            bsum[0] += a[0][s];
            bsum[1] += a[1][s];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the waves' blocks summed in wave order
    __syncthreads();
    float *red = dw_lds;                                        // [wave][4 blocks][16 * 64]
    float *redb = dw_lds + DW_WAVES * 4 * 1024;               // [wave][2][64]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wv * 4 + q) * 1024 + r * 64 + lane] = acc[q][r];
    redb[wv * 128 + lane] = bsum[0];
    redb[wv * 128 + 64 + lane] = bsum[1];
    __syncthreads();
    for (int e = tid; e < 4 * 1024; e += 64 * DW_WAVES) {
        const int q = e >> 10, f = e & 1023, r = f >> 6, l = f & 63;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < DW_WAVES; ++w) s += red[(4 * w + q) * 1024 + f];
        const int o = o0 + 32 * (q >> 1) + ch_of(r, l >> 5), cc = c0 + 32 * (q & 1) + (l & 31);
        if (o < M && cc < K) out[(long long)o * K + cc] = s;
    }
    if (cs == 0 && tid < 64) {
        const int q = tid >> 5, oc = tid & 31;
        float s = 0.f;
        for (int w = 0; w < DW_WAVES; ++w) s += redb[w * 128 + q * 64 + oc] + redb[w * 128 + q * 64 + 32 + oc];
        const int o = o0 + 32 * q + oc;
        if (o < M) out[(long long)M * K + o] = s;
    }
}
__global__ __launch_bounds__(64 * DW_WAVES) void conv1x1_dw_kernel(
    const float *dyp, long long d_gs, long long d_ld, const float *mp, const float *xp, long long x_gs, long long x_ld,
    const int *nvalid, int G, int N, int M, int K, float *wpart, int tpg, int ntiles) {
    conv1x1_dw_body(dyp, d_gs, d_ld, mp, xp, x_gs, x_ld, nvalid, G, N, M, K, wpart + blockIdx.x * ((long long)M * K + M),
                    blockIdx.y, tpg, ntiles);
}
// several layers in one launch (the three convs of an MlpBlock_Real): blockIdx.y walks the sets of all jobs; a chunk's
// record is the concatenation of the jobs' records, so ONE fgnn_reduce_partials finishes all of them
struct DwJobs {
    const float *dy[FGNN_DW_MAX_JOBS], *mask[FGNN_DW_MAX_JOBS], *x[FGNN_DW_MAX_JOBS];
    long long d_gs[FGNN_DW_MAX_JOBS], d_ld[FGNN_DW_MAX_JOBS], x_gs[FGNN_DW_MAX_JOBS], x_ld[FGNN_DW_MAX_JOBS];
    int M[FGNN_DW_MAX_JOBS], K[FGNN_DW_MAX_JOBS], set0[FGNN_DW_MAX_JOBS + 1];
    long long off[FGNN_DW_MAX_JOBS + 1];
};
__global__ __launch_bounds__(64 * DW_WAVES) void conv1x1_dw_multi_kernel(const DwJobs J, int njobs, const int *nvalid, int G, int N,
                                                                         float *wpart, int tpg, int ntiles) {
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.y >= J.set0[j + 1]) ++j;
    conv1x1_dw_body(J.dy[j], J.d_gs[j], J.d_ld[j], J.mask[j], J.x[j], J.x_gs[j], J.x_ld[j], nvalid, G, N, J.M[j], J.K[j],
                    wpart + blockIdx.x * J.off[njobs] + J.off[j], blockIdx.y - J.set0[j], tpg, ntiles);
}
constexpr int DW_LDS_BYTES = DW_WAVES * DW_TILES * 32 * CONV_TLD * 4;
static_assert(DW_LDS_BYTES >= DW_WAVES * 4 * 1024 * 4 + DW_WAVES * 128 * 4, "dw LDS: the reduction image must fit");

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
    const int K2 = (K + 1) / 2, MG = (M + 31) / 32;

    if (K2 <= 64) {         // all input channels of a tile in registers
        const size_t lds = ((size_t)MG * K2 * 64 + MG * 32) * sizeof(float);
#define FGNN_CONV_PRE(KH)                                                                                              \
    {                                                                                                                  \
        static LdsAttrCache attr_cache;                                                                                \
        FGNN_CHECK(fgnn_raise_lds(attr_cache, (const void *)conv1x1_pre_kernel<KH>, lds), "fgnn_conv1x1: %zu bytes of LDS refused", lds); \
        hipLaunchKernelGGL(conv1x1_pre_kernel<KH>, dim3(grid), dim3(64 * CONV_WAVES), lds, (hipStream_t)stream, x,      \
                           x_gstride, x_ld, relu_mask, W, w_ostride, w_kstride, bias, relu, nvalid, G, N, M, K, y,     \
                           y_gstride, y_ld, tpg, ntiles);                                                              \
    }
        if (K2 <= 4) FGNN_CONV_PRE(4)
        else if (K2 <= 16) FGNN_CONV_PRE(16)
        else if (K2 <= 32) FGNN_CONV_PRE(32)
        else FGNN_CONV_PRE(64)
#undef FGNN_CONV_PRE
        FGNN_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = (size_t)K2 * 64 * sizeof(float);
    hipLaunchKernelGGL(conv1x1_kernel, dim3(grid), dim3(64 * CONV_WAVES), lds, (hipStream_t)stream, x, x_gstride, x_ld,
                       relu_mask, W, w_ostride, w_kstride, bias, relu, nvalid, G, N, M, K, y, y_gstride, y_ld, tpg, ntiles);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_conv1x1_dw_chunks(int G, int N) {
    int tpg;
    const int ntiles = conv_tiles(G, N, &tpg);
    int chunks = (ntiles + 4 * CONV_WAVES - 1) / (4 * CONV_WAVES);
    return chunks > 256 ? 256 : (chunks < 1 ? 1 : chunks);      // one workgroup per CU and set of blocks
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
    const int sets = (((M + 31) / 32 + 1) / 2) * (((K + 31) / 32 + 1) / 2);
    static LdsAttrCache attr_cache;
    FGNN_CHECK(fgnn_raise_lds(attr_cache, (const void *)conv1x1_dw_kernel, DW_LDS_BYTES), "fgnn_conv1x1_dw: %d bytes of LDS refused", DW_LDS_BYTES);
    hipLaunchKernelGGL(conv1x1_dw_kernel, dim3(chunks, sets), dim3(64 * DW_WAVES), DW_LDS_BYTES, (hipStream_t)stream, dy,
                       d_gstride, d_ld, relu_mask, x, x_gstride, x_ld, nvalid, G, N, M, K, wpart, tpg, ntiles);
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_conv_chain_supported(int depth, int K0, const int *M) {
    if (depth < 1 || depth > 3 || K0 < 1 || K0 > 128 || !M) return 0;
    for (int l = 0; l < depth; ++l) {
        if (M[l] < 1 || M[l] > 128) return 0;
        if (l < depth - 1 && M[l] > 64) return 0;       // hidden activations: at most two fragment groups feed the next layer
    }
    return 1;
}

extern "C" int fgnn_conv_chain(const fgnn_chain_args *a, void *stream) {
    FGNN_CHECK(a && a->x && a->G > 0 && a->N > 0, "fgnn_conv_chain: bad arguments");
    int Ms[3] = {0, 0, 0};
    for (int l = 0; l < a->depth && l < 3; ++l) Ms[l] = a->layer[l].M;
    FGNN_CHECK(fgnn_conv_chain_supported(a->depth, a->layer[0].K, Ms),
               "fgnn_conv_chain: depth %d, widths %d -> %d -> %d -> %d are outside the chain kernel (use fgnn_conv1x1 per layer)",
               a->depth, a->layer[0].K, Ms[0], Ms[1], Ms[2]);
    const long long P = (long long)a->N * a->N, lim = (1ll << 31) / 4;
    FGNN_CHECK(a->x_ld >= P && a->x_gstride >= a->layer[0].K * a->x_ld && a->G * a->x_gstride < lim,
               "fgnn_conv_chain: input strides smaller than the tensor, or 2 GiB or more");
    ChainDev L = {};
    L.o_ld = a->o_ld;
    size_t floats = 0;
    int maxg = 1;
    for (int l = 0; l < a->depth; ++l) {
        const fgnn_chain_layer &y = a->layer[l];
        FGNN_CHECK(y.W && y.M > 0 && y.K == (l == 0 ? y.K : a->layer[l - 1].M), "fgnn_conv_chain: layer %d does not chain (K = %d)", l, y.K);
        FGNN_CHECK(!(y.out || y.mask) || (a->o_ld >= P && y.o_gstride >= y.M * a->o_ld && a->G * y.o_gstride < lim),
                   "fgnn_conv_chain: layer %d output strides smaller than the tensor, or 2 GiB or more", l);
        L.W[l] = y.W; L.w_so[l] = y.w_ostride; L.w_sk[l] = y.w_kstride; L.bias[l] = y.bias; L.M[l] = y.M; L.K[l] = y.K;
        L.relu[l] = y.relu; L.mask[l] = y.mask; L.out[l] = y.out; L.o_gs[l] = y.o_gstride;
        const int og = (y.M + 31) / 32;
        maxg = og > maxg ? og : maxg;
        floats += l == 0 ? (size_t)og * ((y.K + 1) / 2) * 64 : (size_t)og * ((a->layer[l - 1].M + 31) / 32) * 1024;
    }
    FGNN_CHECK(a->layer[a->depth - 1].out, "fgnn_conv_chain: the last layer needs an output");
    const size_t lds = (floats + 3 * 128) * sizeof(float);
    FGNN_CHECK(lds <= 160 * 1024, "fgnn_conv_chain: %zu bytes of operand images exceed the LDS", lds);
    int tpg;
    const int ntiles = conv_tiles(a->G, a->N, &tpg);
    int grid = (ntiles + 2 * CONV_WAVES - 1) / (2 * CONV_WAVES);
    if (grid > 1024) grid = 1024;
    const int K2 = (a->layer[0].K + 1) / 2;
#define FGNN_CHAIN_(KH, MG, FL)                                                                                               \
    {                                                                                                                      \
        static LdsAttrCache attr_cache;                                                                                \
        FGNN_CHECK(fgnn_raise_lds(attr_cache, (const void *)conv_chain_kernel<KH, MG, FL>, lds), "fgnn_conv_chain: %zu bytes of LDS refused", lds); \
        hipLaunchKernelGGL((conv_chain_kernel<KH, MG, FL>), dim3(grid), dim3(64 * CONV_WAVES), lds, (hipStream_t)stream, a->x,  \
                           a->x_gstride, a->x_ld, a->depth, L, a->nvalid, a->G, a->N, tpg, ntiles);                        \
    }
    // FULL instantiation: every layer has exactly MG full output groups -- the MG the kernel is INSTANTIATED with (2 or 4), not the
    // largest group count of this chain -- and K_0 = 2 KH for the instantiated KH.  (Until round 5 this compared against `maxg`: a
    // chain of 32-wide layers with K_0 in {8, 32, 64, 128} -- the input-gradient chain of every MlpBlock_Real(> 64 -> 32), the forward
    // of MlpBlock_Real(128 -> 32) -- ran the two-group kernel on one-group layers: wrong gradients, found by tests/diag/gpu_fuzz_models.py
    // with another seed; tests/test_gpu_widths.py::test_mlp_block_width_sweep now covers these shapes.)
    const int mg_inst = maxg <= 2 ? 2 : 4;
    bool full_all = true;
    for (int l = 0; l < a->depth; ++l) full_all = full_all && a->layer[l].M == 32 * mg_inst;
#define FGNN_CHAIN(KH, MG)                                      \
    {                                                           \
        if (full_all && a->layer[0].K == 2 * KH) FGNN_CHAIN_(KH, MG, true) \
        else FGNN_CHAIN_(KH, MG, false)                         \
    }
    if (maxg <= 2) {
        if (K2 <= 4) FGNN_CHAIN(4, 2)
        else if (K2 <= 16) FGNN_CHAIN(16, 2)
        else if (K2 <= 32) FGNN_CHAIN(32, 2)
        else FGNN_CHAIN(64, 2)
    } else {
        if (K2 <= 16) FGNN_CHAIN(16, 4)
        else if (K2 <= 32) FGNN_CHAIN(32, 4)
        else FGNN_CHAIN(64, 4)
    }
#undef FGNN_CHAIN
#undef FGNN_CHAIN_
    FGNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int fgnn_conv1x1_dw_multi(const fgnn_dw_job *jobs, int njobs, const int *nvalid, int G, int N, float *wpart,
                                     void *stream) {
    FGNN_CHECK(jobs && wpart && njobs >= 1 && njobs <= FGNN_DW_MAX_JOBS && G > 0 && N > 0, "fgnn_conv1x1_dw_multi: bad arguments");
    const long long P = (long long)N * N, lim = (1ll << 31) / 4;
    DwJobs J = {};
    int sets = 0;
    long long off = 0;
    for (int j = 0; j < njobs; ++j) {
        const fgnn_dw_job &b = jobs[j];
        FGNN_CHECK(b.dy && b.x && b.M > 0 && b.K > 0 && b.M <= FGNN_CONV_MAX_CH && b.K <= FGNN_CONV_MAX_CH,
                   "fgnn_conv1x1_dw_multi: job %d: bad tensors or more than %d channels", j, FGNN_CONV_MAX_CH);
        FGNN_CHECK(b.x_ld >= P && b.d_ld >= P && b.x_gstride >= b.K * b.x_ld && b.d_gstride >= b.M * b.d_ld &&
                       G * b.x_gstride < lim && G * b.d_gstride < lim,
                   "fgnn_conv1x1_dw_multi: job %d: strides smaller than the tensors, or 2 GiB or more", j);
        J.dy[j] = b.dy; J.mask[j] = b.relu_mask; J.x[j] = b.x;
        J.d_gs[j] = b.d_gstride; J.d_ld[j] = b.d_ld; J.x_gs[j] = b.x_gstride; J.x_ld[j] = b.x_ld;
        J.M[j] = b.M; J.K[j] = b.K; J.set0[j] = sets; J.off[j] = off;
        sets += (((b.M + 31) / 32 + 1) / 2) * (((b.K + 31) / 32 + 1) / 2);
        off += (long long)b.M * b.K + b.M;
    }
    J.set0[njobs] = sets;
    J.off[njobs] = off;
    FGNN_CHECK((long long)G * ((P + CONV_T - 1) / CONV_T) < (1ll << 30), "fgnn_conv1x1_dw_multi: too many tiles");
    int tpg;
    const int ntiles = conv_tiles(G, N, &tpg);
    const int chunks = fgnn_conv1x1_dw_chunks(G, N);
    static LdsAttrCache attr_cache;
    FGNN_CHECK(fgnn_raise_lds(attr_cache, (const void *)conv1x1_dw_multi_kernel, DW_LDS_BYTES), "fgnn_conv1x1_dw_multi: %d bytes of LDS refused", DW_LDS_BYTES);
    hipLaunchKernelGGL(conv1x1_dw_multi_kernel, dim3(chunks, sets), dim3(64 * DW_WAVES), DW_LDS_BYTES, (hipStream_t)stream, J, njobs,
                       nvalid, G, N, wpart, tpg, ntiles);
    FGNN_LAUNCH_CHECK();
    return 0;
}
