// Fused MlpBlock_Real forward (conv1x1 + ReLU chain, last conv linear) for gfx950.
//
// Replaces models/layers.py:126-131 of the reference (the GraphNorm reductions of
// :72-73 are emitted as per-tile partials; the normalisation itself is applied by the
// consumer on load).  Formulation: out[o][p] = sum_c W[o][c] in[c][p] with the 32 output
// channels as MFMA rows and 32 pixels as MFMA columns, v_mfma_f32_32x32x2_f32 (exact
// fp32).  The D fragment of one layer (lane = pixel, register = channel) is directly the
// B operand of the next layer, so the whole conv/ReLU chain stays in registers; HBM sees
// one read of the input slabs and one write of z per MLP.
//
// Work decomposition: a tile = 32 consecutive pixels of one graph (all channels).  Waves
// are persistent (one per SIMD, 256 workgroups x 4 waves), each takes a contiguous tile
// range and processes two tiles of one graph at a time (two independent MFMA chains hide
// the MFMA->VALU->MFMA gaps); raw inputs of the next tiles are prefetched into registers
// while the current ones compute.  The per-graph GraphNorm records of the input slabs are
// cached in registers and (re)loaded BEFORE the prefetch is issued, so that the in-order
// vmcnt wait for them does not drain the prefetch.
// Tile statistics: the z tile is transposed through a wave-private LDS tile so that every
// lane owns one channel and sums its 16 pixels in registers (two-pass mean / M2).
#include "fgnn_common.h"

namespace {

constexpr int TLD = 36;              // LDS tile row stride (floats)
constexpr int TILE_F = 32 * TLD;

struct TileCtx {
    int g, tt, p;
    bool inb, valid;
};

DEVI TileCtx decode_tile(int tile, bool active, int tpg, int N, int P, const int *nvalid, int j) {
    TileCtx c;
    c.g = __builtin_amdgcn_readfirstlane(active ? tile / tpg : 0);
    c.tt = active ? tile - c.g * tpg : 0;
    c.p = c.tt * FGNN_TILE + j;
    c.inb = active && c.p < P;
    const int i = c.p / N;
    const int jj = c.p - i * N;
    const int nv = nvalid_of(nvalid, c.g, N);
    c.valid = c.inb && i < nv && jj < nv;
    return c;
}

// channel contracted by k-step k in half-wave h for a slab with S = C/2 steps: 32-channel
// slabs use the accumulator-fragment pairing (same as the hidden layers), narrower ones (2k, 2k+1)
template <int S>
DEVI constexpr int slab_ch(int k, int h) { return S == 16 ? ch_of(k, h) : 2 * k + h; }

template <int S>
DEVI constexpr int slab_kbase(int k) { return S == 16 ? (k & 3) + 8 * (k >> 2) : 2 * k; }
template <int S>
DEVI constexpr int slab_hmul() { return S == 16 ? 4 : 1; }

template <int HMUL>
DEVI int lane_off(const View &v, const TileCtx &c, int h) {
    return c.inb ? HMUL * h * v.ld4 + 4 * c.p : OOB_OFF;
}

template <int S>
DEVI void load_raw(float (&x)[S > 0 ? S : 1], const View &v, const TileCtx &c, int h) {
    if constexpr (S > 0) {
        const int voff = lane_off<slab_hmul<S>()>(v, c, h);
        const int s0 = c.g * v.gs4;
#pragma unroll
        for (int k = 0; k < S; ++k) x[k] = buf_load(v, voff, s0 + slab_kbase<S>(k) * v.ld4);
    }
}

template <int S>
struct NormCache {
    float mean[S > 0 ? S : 1], a[S > 0 ? S : 1], beta[S > 0 ? S : 1];
};

template <int S>
DEVI void load_norm(NormCache<S> &nc, const fgnn_slab &s, int g, int h) {
    if constexpr (S > 0) {
        if (s.nrm) {
            const float4 *nr = reinterpret_cast<const float4 *>(s.nrm) + (long long)g * s.C;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int ch = slab_ch<S>(k, h);
                const float4 n = nr[ch];
                nc.mean[k] = n.x;
                nc.a[k] = n.y;
                nc.beta[k] = s.beta ? s.beta[ch] : 0.f;
            }
        }
    }
}

template <int S>
DEVI void apply_norm(float (&x)[S > 0 ? S : 1], const NormCache<S> &nc, bool on, bool valid) {
    if constexpr (S > 0) {
        if (on) {
#pragma unroll
            for (int k = 0; k < S; ++k) x[k] = valid ? (x[k] - nc.mean[k]) * nc.a[k] + nc.beta[k] : 0.f;
        }
    }
}

template <int CA, int CB, int NMLP, int DEPTH>
struct FwdWeights {
    static constexpr int SA = CA / 2, SB = CB / 2;
    float w1a[NMLP][SA > 0 ? SA : 1];
    float w1b[NMLP][SB > 0 ? SB : 1];
    float wh[NMLP][DEPTH > 1 ? DEPTH - 1 : 1][16];
    float bv[NMLP][DEPTH][16];
};

template <int CA, int CB, int NMLP, int DEPTH, int NT>
DEVI void fwd_compute(const fgnn_mlp_fwd_args &A, const FwdWeights<CA, CB, NMLP, DEPTH> &w,
                      const NormCache<CA / 2> &nca, const NormCache<CB / 2> &ncb, const int tile0,
                      float (&xa)[2][CA / 2 > 0 ? CA / 2 : 1], float (&xb)[2][CB / 2 > 0 ? CB / 2 : 1],
                      const View (&vz)[NMLP], float *lds, int tpg, int P, int lane) {
    constexpr int SA = CA / 2, SB = CB / 2;
    const int j = lane & 31, h = lane >> 5;
    TileCtx c[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        c[t] = decode_tile(tile0 + t, true, tpg, A.N, P, A.nvalid, j);
        apply_norm<SA>(xa[t], nca, A.a.nrm != nullptr, c[t].valid);
        apply_norm<SB>(xb[t], ncb, A.b.nrm != nullptr, c[t].valid);
    }
#pragma unroll
    for (int m = 0; m < NMLP; ++m) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = w.bv[m][0][r];
#pragma unroll
        for (int s = 0; s < SA; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(w.w1a[m][s], xa[t][s], acc[t]);
#pragma unroll
        for (int s = 0; s < SB; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32(w.w1b[m][s], xb[t][s], acc[t]);
#pragma unroll
        for (int l = 1; l < DEPTH; ++l) {
            float hid[NT][16];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hid[t][r] = fmaxf(acc[t][r], 0.f);
                    acc[t][r] = w.bv[m][l][r];
                }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma32(w.wh[m][l - 1][r], hid[t][r], acc[t]);
        }
        // epilogue: mask, store z, transpose through LDS, per-tile {mean, M2} with lane = channel
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned vmask = (unsigned)__ballot(c[t].valid);      // bit px = pixel valid (low half-wave)
            const float cnt = (float)__popc(vmask);
            const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
            const int zoff = lane_off<4>(vz[m], c[t], h);
            const int zs0 = c[t].g * vz[m].gs4;
            float *tl = lds + t * TILE_F;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int chl = (r & 3) + 8 * (r >> 2);   // channel minus 4*h
                const float v = c[t].valid ? acc[t][r] : 0.f;
                buf_store(v, vz[m], zoff, zs0 + chl * vz[m].ld4);
                tl[(chl + 4 * h) * TLD + j] = v;
            }
            // lane (ch = j, h) owns pixels 16h .. 16h+15 of channel ch
            const float4 *rp = reinterpret_cast<const float4 *>(tl + j * TLD + 16 * h);
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = rp[k];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (q[k].x + q[k].y) + (q[k].z + q[k].w);
            s += __shfl_xor(s, 32);
            const float mean = s * inv;
            const unsigned mh = vmask >> (16 * h);
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = ((mh >> (4 * k + 0)) & 1u) ? q[k].x - mean : 0.f;
                const float d1 = ((mh >> (4 * k + 1)) & 1u) ? q[k].y - mean : 0.f;
                const float d2 = ((mh >> (4 * k + 2)) & 1u) ? q[k].z - mean : 0.f;
                const float d3 = ((mh >> (4 * k + 3)) & 1u) ? q[k].w - mean : 0.f;
                m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            m2 += __shfl_xor(m2, 32);
            if (h == 0) {
                float2 o;
                o.x = mean;
                o.y = m2;
                reinterpret_cast<float2 *>(A.part[m])[((long long)c[t].g * tpg + c[t].tt) * FGNN_H + j] = o;
            }
            if (m == 0 && lane == 0) A.cnt[(long long)c[t].g * tpg + c[t].tt] = cnt;
        }
    }
}

template <int CA, int CB, int NMLP, int DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(const fgnn_mlp_fwd_args A, const int tpg,
                                                         const int total_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * TILE_F];
    constexpr int CIN = CA + CB, SA = CA / 2, SB = CB / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + wv;
    const int nwaves = gridDim.x * 4;
    const int P = A.N * A.N;
    float *lds = smem + wv * (2 * TILE_F);
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vb = make_view(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    View vz[NMLP];
#pragma unroll
    for (int m = 0; m < NMLP; ++m) vz[m] = make_view(A.z[m], FGNN_H * A.ldz, A.ldz, A.G);

    FwdWeights<CA, CB, NMLP, DEPTH> w;
#pragma unroll
    for (int m = 0; m < NMLP; ++m) {
#pragma unroll
        for (int s = 0; s < SA; ++s) w.w1a[m][s] = A.W[m][0][o * CIN + slab_ch<SA>(s, h)];
#pragma unroll
        for (int s = 0; s < SB; ++s) w.w1b[m][s] = A.W[m][0][o * CIN + CA + slab_ch<SB>(s, h)];
#pragma unroll
        for (int l = 1; l < DEPTH; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) w.wh[m][l - 1][r] = A.W[m][l][o * FGNN_H + ch_of(r, h)];
#pragma unroll
        for (int l = 0; l < DEPTH; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) w.bv[m][l][r] = A.bias[m][l][ch_of(r, h)];
    }

    const int q = total_tiles / nwaves, rem = total_tiles % nwaves;
    const int t0 = wave * q + (wave < rem ? wave : rem);
    const int t1 = t0 + q + (wave < rem ? 1 : 0);

    NormCache<SA> nca;
    NormCache<SB> ncb;
    int cached_g = -1;

    float xa[2][SA > 0 ? SA : 1], xb[2][SB > 0 ? SB : 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const TileCtx c = decode_tile(t0 + t, t0 + t < t1, tpg, A.N, P, A.nvalid, o);
        load_raw<SA>(xa[t], va, c, h);
        load_raw<SB>(xb[t], vb, c, h);
    }
    int tile = t0;
    while (tile < t1) {
        const int g = __builtin_amdgcn_readfirstlane(tile / tpg);
        const bool pair = (tile + 1 < t1) && ((tile + 1) / tpg == g);
        const int step = pair ? 2 : 1;
        if (g != cached_g) {          // wave-uniform; issued before the prefetch (vmcnt is in-order)
            load_norm<SA>(nca, A.a, g, h);
            load_norm<SB>(ncb, A.b, g, h);
            cached_g = g;
        }
        float na[2][SA > 0 ? SA : 1], nb[2][SB > 0 ? SB : 1];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nt = tile + step + t;
            const TileCtx c = decode_tile(nt, nt < t1, tpg, A.N, P, A.nvalid, o);
            load_raw<SA>(na[t], va, c, h);
            load_raw<SB>(nb[t], vb, c, h);
        }
        if (pair)
            fwd_compute<CA, CB, NMLP, DEPTH, 2>(A, w, nca, ncb, tile, xa, xb, vz, lds, tpg, P, lane);
        else
            fwd_compute<CA, CB, NMLP, DEPTH, 1>(A, w, nca, ncb, tile, xa, xb, vz, lds, tpg, P, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < SA; ++s) xa[t][s] = na[t][s];
#pragma unroll
            for (int s = 0; s < SB; ++s) xb[t][s] = nb[t][s];
        }
        tile += step;
    }
}

template <int CA, int CB, int NMLP, int DEPTH>
int launch_fwd(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st) {
    int nwaves = total < 1024 ? total : 1024;
    int grid = (nwaves + 3) / 4;
    hipLaunchKernelGGL((mlp_fwd_kernel<CA, CB, NMLP, DEPTH>), dim3(grid), dim3(256), 0, st, *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}

template <int NMLP, int DEPTH>
int dispatch_c(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st) {
    const int ca = a->a.C, cb = a->b.C;
#define FGNN_CASE(A_, B_) \
    if (ca == A_ && cb == B_) return launch_fwd<A_, B_, NMLP, DEPTH>(a, tpg, total, st);
    FGNN_CASE(2, 0)
    FGNN_CASE(16, 0)
    FGNN_CASE(32, 0)
    if constexpr (NMLP == 1) {      // two-slab inputs (mlp3 = [mult ; in]) only come with a single MLP
        FGNN_CASE(32, 2)
        FGNN_CASE(32, 32)
    }
#undef FGNN_CASE
    fgnn_set_error("fgnn_mlp_fwd: unsupported input channels (%d + %d) for nmlp=%d; built for 2, 16, 32 and, with nmlp=1, 32+2, 32+32", ca, cb, NMLP);
    return 1;
}

}  // namespace

extern "C" int fgnn_tiles_per_graph(int N) { return (N * N + FGNN_TILE - 1) / FGNN_TILE; }

extern "C" int fgnn_mlp_fwd(const fgnn_mlp_fwd_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_fwd: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_fwd: bad G=%d N=%d", a->G, a->N);
    FGNN_CHECK(a->nmlp == 1 || a->nmlp == 2, "fgnn_mlp_fwd: nmlp must be 1 or 2 (got %d)", a->nmlp);
    FGNN_CHECK(a->depth >= 1 && a->depth <= FGNN_MAX_DEPTH, "fgnn_mlp_fwd: depth %d not in 1..%d", a->depth, FGNN_MAX_DEPTH);
    FGNN_CHECK(a->a.ptr && a->a.C > 0, "fgnn_mlp_fwd: slab a missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr, "fgnn_mlp_fwd: slab b has channels but no pointer");
    FGNN_CHECK((long long)a->N * a->N <= a->ldz && (long long)a->N * a->N <= a->a.ldp, "fgnn_mlp_fwd: channel stride < N*N");
    for (int m = 0; m < a->nmlp; ++m) {
        FGNN_CHECK(a->z[m] && a->part[m], "fgnn_mlp_fwd: missing output %d", m);
        for (int l = 0; l < a->depth; ++l) FGNN_CHECK(a->W[m][l] && a->bias[m][l], "fgnn_mlp_fwd: missing weights mlp %d layer %d", m, l);
    }
    FGNN_CHECK(a->cnt, "fgnn_mlp_fwd: missing cnt");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * FGNN_H * a->ldz < lim,
                   "fgnn_mlp_fwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_fwd: too many tiles");
    hipStream_t st = (hipStream_t)stream;
#define FGNN_ND(NM_, D_) \
    if (a->nmlp == NM_ && a->depth == D_) return dispatch_c<NM_, D_>(a, tpg, (int)total, st);
    FGNN_ND(1, 1) FGNN_ND(1, 2) FGNN_ND(1, 3) FGNN_ND(2, 1) FGNN_ND(2, 2) FGNN_ND(2, 3)
#undef FGNN_ND
    fgnn_set_error("fgnn_mlp_fwd: unsupported nmlp/depth");
    return 1;
}
