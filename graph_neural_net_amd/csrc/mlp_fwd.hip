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
// range and processes two tiles at a time (two independent MFMA chains hide the
// MFMA->VALU->MFMA gaps); raw inputs of the next pair are prefetched into registers
// while the current pair computes.
#include "fgnn_common.h"

namespace {

struct TileCtx {
    int g, tt, p;
    bool inb, valid;
};

DEVI TileCtx decode_tile(int tile, bool active, int tpg, int N, int P, const int *nvalid, int j) {
    TileCtx c;
    c.g = active ? tile / tpg : 0;
    c.tt = active ? tile - c.g * tpg : 0;
    c.p = c.tt * FGNN_TILE + j;
    c.inb = active && c.p < P;
    const int i = c.p / N;
    const int jj = c.p - i * N;
    const int nv = nvalid_of(nvalid, c.g, N);
    c.valid = c.inb && i < nv && jj < nv;
    return c;
}

template <int S>
DEVI void load_raw(float (&x)[S > 0 ? S : 1], const fgnn_slab &s, const TileCtx &c, int h) {
    if constexpr (S > 0) {
        const float *base = s.ptr + (long long)c.g * s.gstride + (long long)h * s.ldp + c.p;
#pragma unroll
        for (int k = 0; k < S; ++k) x[k] = c.inb ? base[(long long)(2 * k) * s.ldp] : 0.f;
    }
}

template <int S>
DEVI void apply_norm(float (&x)[S > 0 ? S : 1], const fgnn_slab &s, const TileCtx &c, int h) {
    if constexpr (S > 0) {
        if (s.nrm) {
            const float4 *nr = reinterpret_cast<const float4 *>(s.nrm) + (long long)c.g * s.C + h;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const float4 n = nr[2 * k];
                const float be = s.beta ? s.beta[2 * k + h] : 0.f;
                x[k] = c.valid ? (x[k] - n.x) * n.y + be : 0.f;
            }
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
                      const int (&tiles)[NT], float (&xa)[NT][CA / 2 > 0 ? CA / 2 : 1],
                      float (&xb)[NT][CB / 2 > 0 ? CB / 2 : 1], int tpg, int P, int lane) {
    constexpr int SA = CA / 2, SB = CB / 2;
    const int j = lane & 31, h = lane >> 5;
    TileCtx c[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        c[t] = decode_tile(tiles[t], true, tpg, A.N, P, A.nvalid, j);
        apply_norm<SA>(xa[t], A.a, c[t], h);
        apply_norm<SB>(xb[t], A.b, c[t], h);
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
        // epilogue: mask, store z, per-tile {mean, M2}
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned long long bal = __ballot(c[t].valid);
            const float cnt = (float)__popc((unsigned)bal);
            const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
            float *zp = A.z[m] + ((long long)c[t].g * FGNN_H + 4 * h) * A.ldz + c[t].p;
            float *pp = A.part[m] + (((long long)c[t].g * tpg + c[t].tt) * FGNN_H + 4 * h) * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int chl = (r & 3) + 8 * (r >> 2);   // channel minus 4*h
                const float v = c[t].valid ? acc[t][r] : 0.f;
                if (c[t].inb) zp[(long long)chl * A.ldz] = v;
                const float mean = half_sum(v) * inv;
                const float d = c[t].valid ? v - mean : 0.f;
                const float m2 = half_sum(d * d);
                if (j == 0) {
                    pp[chl * 2 + 0] = mean;
                    pp[chl * 2 + 1] = m2;
                }
            }
            if (m == 0 && lane == 0) A.cnt[(long long)c[t].g * tpg + c[t].tt] = cnt;
        }
    }
}

template <int CA, int CB, int NMLP, int DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(const fgnn_mlp_fwd_args A, const int tpg,
                                                         const int total_tiles) {
    constexpr int CIN = CA + CB, SA = CA / 2, SB = CB / 2;
    const int lane = threadIdx.x & 63;
    const int o = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int P = A.N * A.N;

    FwdWeights<CA, CB, NMLP, DEPTH> w;
#pragma unroll
    for (int m = 0; m < NMLP; ++m) {
#pragma unroll
        for (int s = 0; s < SA; ++s) w.w1a[m][s] = A.W[m][0][o * CIN + 2 * s + h];
#pragma unroll
        for (int s = 0; s < SB; ++s) w.w1b[m][s] = A.W[m][0][o * CIN + CA + 2 * s + h];
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

    float xa[2][SA > 0 ? SA : 1], xb[2][SB > 0 ? SB : 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const TileCtx c = decode_tile(t0 + t, t0 + t < t1, tpg, A.N, P, A.nvalid, o);
        load_raw<SA>(xa[t], A.a, c, h);
        load_raw<SB>(xb[t], A.b, c, h);
    }
    int tile = t0;
    while (tile + 2 <= t1) {
        float na[2][SA > 0 ? SA : 1], nb[2][SB > 0 ? SB : 1];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const TileCtx c = decode_tile(tile + 2 + t, tile + 2 + t < t1, tpg, A.N, P, A.nvalid, o);
            load_raw<SA>(na[t], A.a, c, h);
            load_raw<SB>(nb[t], A.b, c, h);
        }
        const int tiles[2] = {tile, tile + 1};
        fwd_compute<CA, CB, NMLP, DEPTH, 2>(A, w, tiles, xa, xb, tpg, P, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < SA; ++s) xa[t][s] = na[t][s];
#pragma unroll
            for (int s = 0; s < SB; ++s) xb[t][s] = nb[t][s];
        }
        tile += 2;
    }
    if (tile < t1) {
        const int tiles[1] = {tile};
        float ya[1][SA > 0 ? SA : 1], yb[1][SB > 0 ? SB : 1];
#pragma unroll
        for (int s = 0; s < SA; ++s) ya[0][s] = xa[0][s];
#pragma unroll
        for (int s = 0; s < SB; ++s) yb[0][s] = xb[0][s];
        fwd_compute<CA, CB, NMLP, DEPTH, 1>(A, w, tiles, ya, yb, tpg, P, lane);
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
    FGNN_CASE(32, 2)
    FGNN_CASE(32, 32)
#undef FGNN_CASE
    fgnn_set_error("fgnn_mlp_fwd: unsupported input channels (%d + %d); built for 2, 16, 32, 32+2, 32+32", ca, cb);
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
