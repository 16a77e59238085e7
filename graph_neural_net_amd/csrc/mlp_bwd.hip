// Fused MlpBlock_Real backward for gfx950 (autograd of models/layers.py:126-131 plus the
// GraphNorm backward of :68-80 folded into the load of dz).
//
// Per 32-pixel tile (one wave, everything in registers / wave-private LDS):
//   1. load the input slabs (normalising on load), recompute the hidden activations with
//      the same in-register MFMA chain as the forward kernel;
//   2. dz = ca*dy + cb*(z-mean) + cc              (coef from fgnn_gn_bwd_coef);
//   3. for l = depth-1 .. 0:
//        dW_l += dpre_l (x) in_l       MFMA over pixels; both operands are transposed
//                                      through a 32x36 LDS tile (conflict-free b128 reads)
//        db_l += dpre_l
//        d in_l = W_l^T dpre_l          in-register chain again (D fragment -> B operand)
//        dpre_{l-1} = d in_l * [h_{l-1} > 0]
//   4. dx written (or accumulated) per input slab.
// dW/db accumulate in registers over the wave's whole tile range; at the end the four
// waves of a workgroup are summed through LDS and one partial per workgroup is written,
// to be reduced in fixed order by fgnn_reduce_partials (deterministic).
#include "fgnn_common.h"

namespace {

constexpr int TLD = 36;              // LDS tile row stride (floats): 144 B rows, 16-B aligned
constexpr int TILE_F = 32 * TLD;     // floats per 32-row tile
constexpr int BWD_WG = 256;          // persistent workgroups (one per CU)

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

// rows ch_of(r,h) of a (G,32,ld) tensor
DEVI void load_rows16(float (&x)[16], const float *ptr, long long gstride, long long ld, const TileCtx &c, int h) {
    const float *base = ptr + (long long)c.g * gstride + (long long)(4 * h) * ld + c.p;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = c.inb ? base[(long long)((r & 3) + 8 * (r >> 2)) * ld] : 0.f;
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

// dW += Dt (rows = out channel) x In (rows = in channel), contraction over the 32 pixels
DEVI f32x16 wgrad_tile(const float *Dt, const float *In, f32x16 acc, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float4 *dp = reinterpret_cast<const float4 *>(Dt + i * TLD + 4 * h);
    const float4 *ip = reinterpret_cast<const float4 *>(In + i * TLD + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = dp[2 * q];
        const float4 b = ip[2 * q];
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
    }
    return acc;
}

DEVI void zero16(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Operand sets kept in workgroup-shared LDS (one float per lane per k-step, stored as
// [step/4][lane][4] so a ds_read_b128 returns four consecutive k-steps of a lane).
template <int CA, int CB, int DEPTH>
struct BwdLayout {
    static constexpr int SA = CA / 2, SB = CB / 2;
    static constexpr int pad4(int x) { return (x + 3) & ~3; }
    static constexpr int OFF_W1A = 0;                                     // SA steps: forward layer 0, slab a
    static constexpr int OFF_W1B = OFF_W1A + pad4(SA);                    // SB steps: forward layer 0, slab b
    static constexpr int OFF_WH = OFF_W1B + pad4(SB);                     // 16*(DEPTH-2): forward layers 1..DEPTH-2
    static constexpr int OFF_BV = OFF_WH + 16 * (DEPTH > 2 ? DEPTH - 2 : 0);   // 16*(DEPTH-1): biases 0..DEPTH-2
    static constexpr int OFF_WT = OFF_BV + 16 * (DEPTH > 1 ? DEPTH - 1 : 0);   // 16*(DEPTH-1): W_l^T, l=1..DEPTH-1
    static constexpr int OFF_WT0A = OFF_WT + 16 * (DEPTH > 1 ? DEPTH - 1 : 0); // 16: W_0^T slab a
    static constexpr int OFF_WT0B = OFF_WT0A + 16;                        // 16: W_0^T slab b
    static constexpr int NSTEPS = OFF_WT0B + (CB > 0 ? 16 : 0);
    static constexpr int WEIGHT_F = NSTEPS * 64;                          // floats
    static constexpr int NTILES = 1 + (CB > 0 ? 1 : 0) + (DEPTH - 1) + 1;
    static constexpr int PCOUNT = 32 * (CA + CB) + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int TILE_F_ALL = 4 * NTILES * TILE_F;
    static constexpr int RED_F = 4 * PCOUNT;
    static constexpr int LDS_F = WEIGHT_F + (TILE_F_ALL > RED_F ? TILE_F_ALL : RED_F);
};

template <int OFF, int CNT>
DEVI void load_ops(float (&dst)[CNT > 0 ? CNT : 1], const float *wl, int lane) {
    static_assert(OFF % 4 == 0, "operand sets are float4 aligned");
    const float4 *p = reinterpret_cast<const float4 *>(wl) + (OFF / 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < (CNT + 3) / 4; ++q) {
        const float4 v = p[q * 64];
        if (4 * q + 0 < CNT) dst[4 * q + 0] = v.x;
        if (4 * q + 1 < CNT) dst[4 * q + 1] = v.y;
        if (4 * q + 2 < CNT) dst[4 * q + 2] = v.z;
        if (4 * q + 3 < CNT) dst[4 * q + 3] = v.w;
    }
}

template <int CA, int CB, int DEPTH>
__global__ __launch_bounds__(256, 1) void mlp_bwd_kernel(const fgnn_mlp_bwd_args A, const int tpg,
                                                         const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BwdLayout<CA, CB, DEPTH>;
    constexpr int CIN = CA + CB, SA = CA / 2, SB = CB / 2;
    constexpr int NTILES = L::NTILES;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wave = blockIdx.x * 4 + wv;
    const int nwaves = gridDim.x * 4;
    const int P = A.N * A.N;

    float *wl = smem;                                   // shared operand sets
    float *tiles = smem + L::WEIGHT_F;
    float *my = tiles + wv * (NTILES * TILE_F);
    float *XA = my;
    float *XB = my + TILE_F;                                  // only if CB > 0
    float *HT = my + (1 + (CB > 0 ? 1 : 0)) * TILE_F;         // DEPTH-1 tiles
    float *DT = HT + (DEPTH - 1) * TILE_F;
    for (int e = lane; e < NTILES * TILE_F; e += WAVE) my[e] = 0.f;

    // ---- fill the operand sets (float4 groups round-robin over the 4 waves) ----
#define FGNN_PUT(t_, expr_)                                                        \
    do {                                                                           \
        if ((((t_) >> 2) & 3) == wv) wl[((t_) >> 2) * 256 + lane * 4 + ((t_) & 3)] = (expr_); \
    } while (0)
#pragma unroll
    for (int s = 0; s < SA; ++s) FGNN_PUT(L::OFF_W1A + s, A.W[0][j * CIN + 2 * s + h]);
#pragma unroll
    for (int s = 0; s < SB; ++s) FGNN_PUT(L::OFF_W1B + s, A.W[0][j * CIN + CA + 2 * s + h]);
#pragma unroll
    for (int l = 1; l + 1 < DEPTH; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) FGNN_PUT(L::OFF_WH + 16 * (l - 1) + r, A.W[l][j * FGNN_H + ch_of(r, h)]);
#pragma unroll
    for (int l = 0; l + 1 < DEPTH; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) FGNN_PUT(L::OFF_BV + 16 * l + r, A.bias[l][ch_of(r, h)]);
#pragma unroll
    for (int l = 1; l < DEPTH; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) FGNN_PUT(L::OFF_WT + 16 * (l - 1) + r, A.W[l][ch_of(r, h) * FGNN_H + j]);
#pragma unroll
    for (int r = 0; r < 16; ++r) FGNN_PUT(L::OFF_WT0A + r, (j < CA) ? A.W[0][ch_of(r, h) * CIN + j] : 0.f);
    if constexpr (CB > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) FGNN_PUT(L::OFF_WT0B + r, (j < CB) ? A.W[0][ch_of(r, h) * CIN + CA + j] : 0.f);
    }
#undef FGNN_PUT
    __syncthreads();

    // ---- persistent accumulators ----
    f32x16 dW0a, dW0b, dWh[DEPTH > 1 ? DEPTH - 1 : 1];
    float dbacc[DEPTH][16];
    zero16(dW0a);
    zero16(dW0b);
#pragma unroll
    for (int l = 0; l + 1 < DEPTH; ++l) zero16(dWh[l]);
#pragma unroll
    for (int l = 0; l < DEPTH; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) dbacc[l][r] = 0.f;

    const int q = total_tiles / nwaves, rem = total_tiles % nwaves;
    const int t0 = wave * q + (wave < rem ? wave : rem);
    const int t1 = t0 + q + (wave < rem ? 1 : 0);

    float xa[SA > 0 ? SA : 1], xb[SB > 0 ? SB : 1], dyr[16], zr[16];
    {
        const TileCtx c = decode_tile(t0, t0 < t1, tpg, A.N, P, A.nvalid, j);
        load_raw<SA>(xa, A.a, c, h);
        load_raw<SB>(xb, A.b, c, h);
        load_rows16(dyr, A.dy, A.dgstride, A.ldd, c, h);
        load_rows16(zr, A.z, A.zgstride, A.ldz, c, h);
    }

    for (int tile = t0; tile < t1; ++tile) {
        // prefetch the next tile's raw operands
        float nxa[SA > 0 ? SA : 1], nxb[SB > 0 ? SB : 1], ndy[16], nz[16];
        {
            const TileCtx c = decode_tile(tile + 1, tile + 1 < t1, tpg, A.N, P, A.nvalid, j);
            load_raw<SA>(nxa, A.a, c, h);
            load_raw<SB>(nxb, A.b, c, h);
            load_rows16(ndy, A.dy, A.dgstride, A.ldd, c, h);
            load_rows16(nz, A.z, A.zgstride, A.ldz, c, h);
        }
        const TileCtx c = decode_tile(tile, true, tpg, A.N, P, A.nvalid, j);
        apply_norm<SA>(xa, A.a, c, h);
        apply_norm<SB>(xb, A.b, c, h);
#pragma unroll
        for (int s = 0; s < SA; ++s) XA[(2 * s + h) * TLD + j] = xa[s];
#pragma unroll
        for (int s = 0; s < SB; ++s) XB[(2 * s + h) * TLD + j] = xb[s];

        // ---- forward recompute of the hidden activations ----
        float hid[DEPTH > 1 ? DEPTH - 1 : 1][16];
        if constexpr (DEPTH > 1) {
            f32x16 acc;
            {
                float b0[16];
                load_ops<L::OFF_BV, 16>(b0, wl, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = b0[r];
                float w1a[SA > 0 ? SA : 1];
                load_ops<L::OFF_W1A, SA>(w1a, wl, lane);
#pragma unroll
                for (int s = 0; s < SA; ++s) acc = mfma32(w1a[s], xa[s], acc);
                if constexpr (SB > 0) {
                    float w1b[SB > 0 ? SB : 1];
                    load_ops<L::OFF_W1B, SB>(w1b, wl, lane);
#pragma unroll
                    for (int s = 0; s < SB; ++s) acc = mfma32(w1b[s], xb[s], acc);
                }
            }
#pragma unroll
            for (int l = 1; l < DEPTH; ++l) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hid[l - 1][r] = fmaxf(acc[r], 0.f);
                    HT[(l - 1) * TILE_F + ch_of(r, h) * TLD + j] = hid[l - 1][r];
                }
                if (l + 1 < DEPTH) {
                    float bl[16], wl_[16];
                    if (l == 1) { load_ops<L::OFF_BV + 16, 16>(bl, wl, lane); load_ops<L::OFF_WH, 16>(wl_, wl, lane); }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = bl[r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc = mfma32(wl_[r], hid[l - 1][r], acc);
                }
            }
        }

        // ---- dz from (dy, z, coef) ----
        float dpre[16];
        {
            const float4 *kp = reinterpret_cast<const float4 *>(A.coef) + (long long)c.g * FGNN_H + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 k = kp[(r & 3) + 8 * (r >> 2)];
                dpre[r] = c.valid ? k.y * dyr[r] + k.z * (zr[r] - k.x) + k.w : 0.f;
            }
        }

        // ---- backward through the layers ----
#pragma unroll
        for (int l = DEPTH - 1; l >= 0; --l) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                DT[ch_of(r, h) * TLD + j] = dpre[r];
                dbacc[l][r] += dpre[r];
            }
            if (l > 0) {
                dWh[l - 1] = wgrad_tile(DT, HT + (l - 1) * TILE_F, dWh[l - 1], lane);
                float wt[16];
                if (l == 1) load_ops<L::OFF_WT, 16>(wt, wl, lane);
                if (l == 2) load_ops<L::OFF_WT + (DEPTH > 2 ? 16 : 0), 16>(wt, wl, lane);
                f32x16 acc;
                zero16(acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc = mfma32(wt[r], dpre[r], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) dpre[r] = hid[l - 1][r] > 0.f ? acc[r] : 0.f;
            } else {
                dW0a = wgrad_tile(DT, XA, dW0a, lane);
                if constexpr (CB > 0) dW0b = wgrad_tile(DT, XB, dW0b, lane);
                if (A.dxa) {
                    float wt[16];
                    load_ops<L::OFF_WT0A, 16>(wt, wl, lane);
                    f32x16 acc;
                    zero16(acc);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc = mfma32(wt[r], dpre[r], acc);
                    float *op = A.dxa + (long long)c.g * A.dxa_gstride + c.p;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = ch_of(r, h);
                        if (ch < CA && c.inb) {
                            float *o = op + (long long)ch * A.dxa_ld;
                            *o = A.accumulate_a ? *o + acc[r] : acc[r];
                        }
                    }
                }
                if constexpr (CB > 0) {
                    if (A.dxb) {
                        float wt[16];
                        load_ops<L::OFF_WT0B, 16>(wt, wl, lane);
                        f32x16 acc;
                        zero16(acc);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc = mfma32(wt[r], dpre[r], acc);
                        float *op = A.dxb + (long long)c.g * A.dxb_gstride + c.p;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ch = ch_of(r, h);
                            if (ch < CB && c.inb) {
                                float *o = op + (long long)ch * A.dxb_ld;
                                *o = A.accumulate_b ? *o + acc[r] : acc[r];
                            }
                        }
                    }
                }
            }
        }

#pragma unroll
        for (int s = 0; s < SA; ++s) xa[s] = nxa[s];
#pragma unroll
        for (int s = 0; s < SB; ++s) xb[s] = nxb[s];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dyr[r] = ndy[r];
            zr[r] = nz[r];
        }
    }

    // ---- workgroup reduction of the parameter gradients ----
    // layout: [W0 (32*CIN) | b0 (32) | W1 (1024) | b1 (32) | ...]
    constexpr int PCOUNT = L::PCOUNT;
    __syncthreads();                       // everyone done with the tile buffers
    float *red = tiles + wv * PCOUNT;      // aliases the tile region
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ch_of(r, h);
            if (j < CA) red[o * CIN + j] = dW0a[r];
            if (CB > 0 && j < CB) red[o * CIN + CA + j] = dW0b[r];
        }
        int off = 32 * CIN;
#pragma unroll
        for (int l = 0; l < DEPTH; ++l) {
            if (l > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[off + ch_of(r, h) * 32 + j] = dWh[l - 1][r];
                off += 1024;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = half_sum(dbacc[l][r]);
                if (j == 0) red[off + ch_of(r, h)] = s;
            }
            off += 32;
        }
    }
    __syncthreads();
    float *out = A.wpart + (long long)blockIdx.x * PCOUNT;
    for (int e = threadIdx.x; e < PCOUNT; e += 256)
        out[e] = (tiles[e] + tiles[PCOUNT + e]) + (tiles[2 * PCOUNT + e] + tiles[3 * PCOUNT + e]);
}

template <int CA, int CB, int DEPTH>
int launch_bwd(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    constexpr int LDS = BwdLayout<CA, CB, DEPTH>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set && LDS > 64 * 1024) {
        (void)hipFuncSetAttribute((const void *)mlp_bwd_kernel<CA, CB, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((mlp_bwd_kernel<CA, CB, DEPTH>), dim3(BWD_WG), dim3(256), LDS, st, *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}

template <int DEPTH>
int dispatch_c(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    const int ca = a->a.C, cb = a->b.C;
#define FGNN_CASE(A_, B_) \
    if (ca == A_ && cb == B_) return launch_bwd<A_, B_, DEPTH>(a, tpg, total, st);
    FGNN_CASE(2, 0)
    FGNN_CASE(16, 0)
    FGNN_CASE(32, 0)
    FGNN_CASE(32, 2)
    FGNN_CASE(32, 32)
#undef FGNN_CASE
    fgnn_set_error("fgnn_mlp_bwd: unsupported input channels (%d + %d); built for 2, 16, 32, 32+2, 32+32", ca, cb);
    return 1;
}

}  // namespace

extern "C" int fgnn_mlp_bwd_num_workgroups(void) { return BWD_WG; }

extern "C" int fgnn_mlp_param_count(int Cin, int depth) { return 32 * Cin + 32 + (depth - 1) * (32 * 32 + 32); }

extern "C" int fgnn_mlp_bwd(const fgnn_mlp_bwd_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_bwd: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_bwd: bad G=%d N=%d", a->G, a->N);
    FGNN_CHECK(a->depth >= 1 && a->depth <= FGNN_MAX_DEPTH, "fgnn_mlp_bwd: depth %d not in 1..%d", a->depth, FGNN_MAX_DEPTH);
    FGNN_CHECK(a->a.ptr && a->a.C > 0, "fgnn_mlp_bwd: slab a missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr, "fgnn_mlp_bwd: slab b has channels but no pointer");
    FGNN_CHECK(a->dy && a->z && a->coef && a->wpart, "fgnn_mlp_bwd: missing dy/z/coef/wpart");
    for (int l = 0; l < a->depth; ++l) FGNN_CHECK(a->W[l] && a->bias[l], "fgnn_mlp_bwd: missing weights layer %d", l);
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (a->depth == 1) return dispatch_c<1>(a, tpg, (int)total, st);
    if (a->depth == 2) return dispatch_c<2>(a, tpg, (int)total, st);
    return dispatch_c<3>(a, tpg, (int)total, st);
}
