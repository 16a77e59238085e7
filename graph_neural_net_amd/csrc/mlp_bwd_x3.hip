// Fused MlpBlock_Real backward (autograd of models/layers.py:126-131 plus the GraphNorm backward of :68-80 folded into the
// load of dz) with every channel / pixel contraction on the bf16 matrix cores through the exact three-way operand split of
// fgnn_x3.h.  Interface, tile geometry, reductions and results (up to fp32 reassociation noise) are those of mlp_bwd.hip;
// see there for the algorithm.  What differs:
//   * a GEMM is 12 v_mfma_f32_32x32x16_bf16 (384 matrix-pipe cycles that run beside the VALU) instead of 16
//     v_mfma_f32_32x32x2_f32 (1024 cycles that block the SIMD's fp32 VALU): the kernel moves from MFMA-bound to
//     VALU-issue / HBM-bound;
//   * its B operand (the fp32 D fragment of the previous GEMM: lane = pixel, register r <-> channel ch_of(r, h)) is split
//     in registers (5.5 VALU per value); weights come pre-split from the operand image (fgnn_pack_x3_operands);
//   * the weight-gradient GEMMs contract over pixels: both operands are read transposed from the wave's fp32 LDS tiles
//     (lane = channel, 16 pixels per lane) exactly as before and split after the read;
//   * the two-slab kernels (mlp3) hold a 1.5x larger operand image and therefore run on TWO tile slots per wave: the
//     first hidden activation stays in registers until the last layer's weight gradient has been issued.
// Built for the reference's depth 3, input slabs of 2 / 32 / 32+2 / 32+32 channels, constant-size batches.
#include "fgnn_tile.h"
#include "fgnn_pack.h"
#include "fgnn_x3.h"

namespace {

constexpr int BWD_WG = 256;          // persistent workgroups (one per CU), = rows of wpart
// waves per workgroup (template parameter NWT): 8 = 2 per SIMD with <= 256 registers each; 4 = one per SIMD with up to 512
// (256 VGPRs + 256 AGPRs: the weight-gradient accumulators live in AGPRs, nothing spills)

// dz coefficients {mean, ca, cb, cc} of one channel: dz = ca*dy + cb*(z - mean) + cc   
DEVI float4 coef_from_sums(const float4 n, const float2 sv, float nv) {
    const float m = nv * nv;
    float4 k;
    k.x = n.x;
    k.y = n.y;
    k.z = m > 0.f ? -n.y * sv.y * n.w / m : 0.f;
    k.w = m > 0.f ? -n.y * sv.x / m : 0.f;
    return k;
}
DEVI float4 coef_record(const fgnn_mlp_bwd_args &A, int g, int ch) {
    if (A.coef) return reinterpret_cast<const float4 *>(A.coef)[(long long)g * FGNN_H + ch];
    const float4 n = reinterpret_cast<const float4 *>(A.znrm)[(long long)g * FGNN_H + ch];
    const float2 sv = reinterpret_cast<const float2 *>(A.s12)[(long long)g * FGNN_H + ch];
    return coef_from_sums(n, sv, (float)nvalid_of(A.nvalid, g, A.N));
}

template <int CA, int CB, int NWT>
struct BwdX3Layout {
    static constexpr int NW = NWT;
    static constexpr int DEPTH = 3;
    static constexpr PkX3 PK = pkx3_layout(1, CA, CB, DEPTH);
    static constexpr int PD = PK.part_dw;
    static constexpr int OFF_W0A = PK.p.off_w0a, OFF_W0B = PK.p.off_w0b;
    static constexpr int OFF_W1 = PK.p.off_wh;                             // forward layer 1 (2 steps)
    static constexpr int OFF_W2T = PK.p.off_wt, OFF_W1T = PK.p.off_wt + 2; // W_2^T, W_1^T (2 steps each)
    static constexpr int OFF_WT0A = PK.p.off_wt0a, OFF_WT0B = PK.p.off_wt0b;
    static constexpr int STEPS_A = pk16_steps(CA), STEPS_B = pk16_steps(CB);
    static constexpr int BIAS_F = PK.bias_off;
    static constexpr int WEIGHT_F = PK.floats;
    static constexpr bool TWO_SLOT = CB > 0;
    static constexpr int NSLOT = TWO_SLOT ? 2 : 3;
    static constexpr int REC_F = (CB > 0 ? 3 : 2) * 32 * 4;               // per wave: nrm a, (nrm b,) coef
    static constexpr int PCOUNT = 32 * (CA + CB) + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int TILE_F_ALL = NW * NSLOT * TILE_F;
    static constexpr int RED_F = NW * PCOUNT;                             // final reduction reuses the whole allocation
    static constexpr int WGK_F = CB > 0 ? FGNN_BWD_COEF_GRAPHS * 128 : 0; // workgroup cache of dz coefficient records (mlp3)
    static constexpr int MAIN_F = WEIGHT_F + NW * REC_F + TILE_F_ALL + WGK_F;
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

// the three bf16 parts of an input slab as B operand (see mlp_fwd_x3.hip)
template <int S>
DEVI void split_slab(X3 &x, const float (&v)[S > 0 ? S : 1], int h, const F16 &negI) {
    if constexpr (S == 16) {
        split16m(x, v, negI);
    } else if constexpr (S == 1) {
        const float other = __shfl_xor(v[0], 32);
        const float a = h == 0 ? v[0] : 0.f, b = h == 0 ? other : 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int d = 0; d < 8; ++d) x.p[q].d[d] = 0u;
        split_pair(a, b, x.p[0].d[0], x.p[1].d[0], x.p[2].d[0]);
    }
}

// 16 pixels of this lane's channel from a 32x36 tile: values 4q + e <-> pixel 4h + 8q + e
DEVI void read_t16(float (&v)[16], const float *T, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float4 *p = reinterpret_cast<const float4 *>(T + i * TLD + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = p[2 * q];
        v[4 * q + 0] = a.x;
        v[4 * q + 1] = a.y;
        v[4 * q + 2] = a.z;
        v[4 * q + 3] = a.w;
    }
}
// transposed operand of a weight-gradient GEMM (+ the bias gradient: the lane's 16 pixels of its channel)
template <bool WITH_DB>
DEVI void load_wgrad_operand(X3 &x, const float *T, float &db, int lane, const F16 &negI) {
    float v[16];
    read_t16(v, T, lane);
    if (WITH_DB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) db += (v[4 * q] + v[4 * q + 1]) + (v[4 * q + 2] + v[4 * q + 3]);
    }
    split16m(x, v, negI);
}
// dW += Dt (rows = out channel) x In (rows = in channel), contraction over the tile's 32 pixels
DEVI f32x16 wgrad_x3(const float *Dt, const float *In, f32x16 acc, float &db, int lane, const F16 &negI) {
    X3 a, b;
    load_wgrad_operand<true>(a, Dt, db, lane, negI);
    float dummy = 0.f;
    load_wgrad_operand<false>(b, In, dummy, lane, negI);
    return gemm_x3_rr<X3_BWD_TERMS>(acc, a, b);
}

DEVI void stage16(float *T, const float (&v)[16], int j, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) T[ch_of(r, h) * TLD + j] = v[r];
}

template <int CA, int CB, bool PK, int NWT>
__global__ __launch_bounds__(64 * NWT, NWT / 4) void mlp_bwd_x3_kernel(const fgnn_mlp_bwd_args A, const int tpg, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BwdX3Layout<CA, CB, NWT>;
    constexpr int NW = NWT;
    constexpr int DEPTH = 3, CIN = CA + CB, SA = CA / 2, SB = CB / 2;
    constexpr bool TWO = L::TWO_SLOT;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int P = A.N * A.N;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vb = make_view(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    PackedSrc ps = {};
    if constexpr (PK) ps = make_packed_src(A.xbits, A.xdeg, A.G, A.N);
    const View vdy = make_view(A.dy, A.dgstride, A.ldd, A.G);
    const View vz = make_view(A.z, A.zgstride, A.ldz, A.G);
    const View vdxa = make_view(A.dxa, A.dxa_gstride, A.dxa_ld, A.G);
    const View vdxb = make_view(A.dxb, A.dxb_gstride, A.dxb_ld, A.G);

    float *wl = smem;                                   // shared operand image
    float *rec = smem + L::WEIGHT_F + wv * L::REC_F;    // wave-private per-graph records
    float *recA = rec, *recB = rec + 128, *recK = rec + (CB > 0 ? 256 : 128);
    float *tiles = smem + L::WEIGHT_F + NW * L::REC_F;
    float *my = tiles + wv * (L::NSLOT * TILE_F);
    // tile slots.  Three slots (single-slab kernels): S0 = h1 / later x_a, S1 = h2 / dpre_1, S2 = dz / dpre_0.
    // Two slots (two-slab kernels): T0 = h2 / h1 / x_a / x_b, T1 = dz / dpre_1 / dpre_0; h1 waits in registers.
    float *S0 = my, *S1 = my + TILE_F, *S2 = my + (TWO ? 1 : 2) * TILE_F;
    float *SH1 = S0;                  // h1 (three slots: staged at once; two slots: after the layer-2 weight gradient)
    float *SH2 = TWO ? S0 : S1;       // h2
    float *SDZ = TWO ? S1 : S2;       // dz
    float *SD1 = S1;                  // dpre_1
    float *SD0 = TWO ? S1 : S2;       // dpre_0
    float *SX = S0;                   // x_a, then x_b

    // ---- persistent accumulators ----
    constexpr bool VW0 = (CA == 2 && CB == 0);          // 2-channel input: layer-0 weight gradient on the VALU (mlp_bwd.hip)
    float w0v[VW0 ? 32 : 1], b0v[VW0 ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (VW0 ? 32 : 1); ++r) w0v[r] = 0.f;
#pragma unroll
    for (int r = 0; r < (VW0 ? 16 : 1); ++r) b0v[r] = 0.f;
    f32x16 dW0a, dW0b, dWh[2];
    float db[DEPTH];
    zero16f(dW0a);
    zero16f(dW0b);
    zero16f(dWh[0]);
    zero16f(dWh[1]);
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] = 0.f;

    // static, strided tile assignment inside the workgroup's contiguous range (fixed accumulation order)
    const int nwg = gridDim.x;
    const int q = total_tiles / nwg, rem = total_tiles % nwg;
    const int T0 = blockIdx.x * q + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    const int T1 = T0 + q + ((int)blockIdx.x < rem ? 1 : 0);
    const bool normA = A.a.nrm != nullptr, normB = (CB > 0) && A.b.nrm != nullptr;
    const bool emit = (CA == 32) && (CB == 0) && normA && A.dxa != nullptr && A.s12part != nullptr;
    const bool from_tiles = (CB > 0) && A.s12tiles != nullptr;
    float *wgK = tiles + L::TILE_F_ALL;
    const int g0 = T0 / tpg;

    // Prologue = ONE memory round trip: operand image (into registers), first tile, per-graph records
    PkRegs<L::WEIGHT_F / 4, 64 * NW> img;
    pk_load_regs(img, A.packed);
    __builtin_amdgcn_sched_barrier(0);
    float xa[SA > 0 ? SA : 1], xb[SB > 0 ? SB : 1];
    float4 rk = make_float4(0.f, 0.f, 0.f, 0.f), ra = rk, rb = rk;
    int cached_g = -1, cur_nv = A.N;
    const int first = T0 + wv;
    {
        const int t = first;
        const TileCtx c = decode_tile(t, t < T1, tpg, A.N, P, j);
        load_slab<SA, PK>(xa, va, ps, c, h);
        load_slab<SB, PK>(xb, vb, ps, c, h);
        if (t < T1 && lane < 32) {
            if (!from_tiles) rk = coef_record(A, c.g, lane);
            if (normA && lane < CA) {
                ra = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                ra.z = A.a.beta ? A.a.beta[lane] : 0.f;
            }
            if (normB && lane < CB) {
                rb = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)c.g * A.b.C + lane];
                rb.z = A.b.beta ? A.b.beta[lane] : 0.f;
            }
        }
        if (t < T1) {
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
    }
    if (from_tiles) {
        // dz coefficients of the graphs this workgroup touches, from the per-tile sums its consumer left behind
        const int g1 = T1 > T0 ? (T1 - 1) / tpg : g0 - 1;
        float2 *scr = reinterpret_cast<float2 *>(tiles);          // [16 slices][32 channels]; tiles are free here
        const int cc = threadIdx.x & 31, sl = threadIdx.x >> 5;
        for (int g = g0; g <= g1; ++g) {
            float p1 = 0.f, p2 = 0.f;
            constexpr int TS = (64 * NW) / 32, U = 8;
            for (int t0 = sl; t0 < tpg; t0 += TS * U) {
                float2 v[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int t = t0 + TS * k;
                    v[k] = reinterpret_cast<const float2 *>(A.s12tiles)[((long long)g * tpg + (t < tpg ? t : 0)) * FGNN_H + cc];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (t0 + TS * k < tpg) {
                        p1 += v[k].x;
                        p2 += v[k].y;
                    }
                }
            }
            scr[sl * 32 + cc] = make_float2(p1, p2);
            __syncthreads();
            if (threadIdx.x < 32) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < (64 * NW) / 32; ++k) {               // fixed order
                    s1 += scr[k * 32 + cc].x;
                    s2 += scr[k * 32 + cc].y;
                }
                const float2 sv = make_float2(s1, s2);
                if (A.s12_out) reinterpret_cast<float2 *>(A.s12_out)[(long long)g * FGNN_H + cc] = sv;
                const float4 n = reinterpret_cast<const float4 *>(A.znrm)[(long long)g * FGNN_H + cc];
                reinterpret_cast<float4 *>(wgK)[(g - g0) * 32 + cc] = coef_from_sums(n, sv, (float)nvalid_of(A.nvalid, g, A.N));
            }
            __syncthreads();
        }
        if (cached_g >= 0 && lane < 32) rk = reinterpret_cast<const float4 *>(wgK)[(cached_g - g0) * 32 + lane];
    }
    pk_store_regs(wl, img);
    if (lane < 32) {
        reinterpret_cast<float4 *>(recK)[lane] = rk;
        reinterpret_cast<float4 *>(recA)[lane] = ra;
        if constexpr (CB > 0) reinterpret_cast<float4 *>(recB)[lane] = rb;
    }
    __syncthreads();

    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const F16 negI = make_neg_identity(lane);
    int tnext = 0;
    for (int tile = first; tile < T1; tile = tnext) {
        tnext = tile + NW;
        const TileCtx c = decode_tile(tile, true, tpg, A.N, P, j);
        if (c.g != cached_g) {
            // per-graph records -> wave-private LDS, issued (and waited for) BEFORE the loads below
            if (lane < 32) {
                const float4 k4 = from_tiles ? reinterpret_cast<const float4 *>(wgK)[(c.g - g0) * 32 + lane]
                                             : coef_record(A, c.g, lane);
                reinterpret_cast<float4 *>(recK)[lane] = k4;
                if (normA && lane < CA) {
                    float4 n = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)c.g * A.a.C + lane];
                    n.z = A.a.beta ? A.a.beta[lane] : 0.f;
                    reinterpret_cast<float4 *>(recA)[lane] = n;
                }
                if (normB && lane < CB) {
                    float4 n = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)c.g * A.b.C + lane];
                    n.z = A.b.beta ? A.b.beta[lane] : 0.f;
                    reinterpret_cast<float4 *>(recB)[lane] = n;
                }
            }
            cached_g = c.g;
            cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, c.g, A.N));
        }
        const bool c_valid = tile_valid(c, cur_nv);
#ifndef X3_EARLY_RMW
#define X3_EARLY_RMW 0
#endif
        constexpr bool EARLY_RMW = (CB == 0) && X3_EARLY_RMW;       // single-slab kernels prefetch the dx they accumulate into
        float dyr[16], zr[16], old[16];
        const bool rmw = EARLY_RMW && A.dxa != nullptr && A.accumulate_a;

        // ---- forward recompute of the hidden activations ----
        f32x16 acc;
        load_bias16(acc, wl + L::BIAS_F, 0, h);
        {
            float ya[SA > 0 ? SA : 1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
            X3 X;
            split_slab<SA>(X, ya, h, negI);
            acc = gemm_x3<L::STEPS_A, X3_FWD_TERMS>(acc, wl, L::PD, L::OFF_W0A, X, lane);
        }
        if constexpr (CB > 0) {
            float yb[SB > 0 ? SB : 1];
            norm_slab<SB>(yb, xb, recB, normB, c_valid, h);
            X3 X;
            split_slab<SB>(X, yb, h, negI);
            acc = gemm_x3<L::STEPS_B, X3_FWD_TERMS>(acc, wl, L::PD, L::OFF_W0B, X, lane);
        }
        // this tile's dy / z and, when accumulating, the current dx values fly behind the recompute
        load_rows16(dyr, vdy, c, h);
        load_rows16(zr, vz, c, h);
        if constexpr (EARLY_RMW) {
            if (rmw) load_rows16(old, vdxa, c, h);
        }
        float h1r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h1r[r] = relu1(acc[r]);
        if constexpr (!TWO) stage16(SH1, h1r, j, h);
        {
            X3 H;
            split16m(H, h1r, negI);
            load_bias16(acc, wl + L::BIAS_F, 1, h);
            acc = gemm_x3<2, X3_FWD_TERMS>(acc, wl, L::PD, L::OFF_W1, H, lane);
        }
        {
            float h2r[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) h2r[r] = relu1(acc[r]);
            stage16(SH2, h2r, j, h);
        }
        // ---- dz from (dy, z, coef) ----
        float dpre[16];
        {
            const float4 *kp = reinterpret_cast<const float4 *>(recK) + 4 * h;
            const float vf = c_valid ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 k = kp[(r & 3) + 8 * (r >> 2)];
                dpre[r] = (k.y * dyr[r] + k.z * (zr[r] - k.x) + k.w) * vf;
            }
        }
        stage16(SDZ, dpre, j, h);
        // ---- layer 2: dgrad (W_2^T dz), weight gradient dz (x) h2, mask with h2 ----
        {
            X3 D;
            split16m(D, dpre, negI);
            acc = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_W2T, D, lane);
        }
        dWh[1] = wgrad_x3(SDZ, SH2, dWh[1], db[2], lane, negI);
        {
            float hsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hsv[r] = SH2[ch_of(r, h) * TLD + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = hsv[r] > 0.f ? acc[r] : 0.f;
        }
        // dpre_1 replaces h2 (three slots) / dz (two slots); LDS is in order within a wave, the readers were issued above
        stage16(SD1, dpre, j, h);
        if constexpr (TWO) stage16(SH1, h1r, j, h);          // h1 takes the slot h2 has just left
        // ---- layer 1 ----
#ifndef X3_RELOAD_X
#define X3_RELOAD_X 0
#endif
        // The 32-channel input slab is not held in registers across the tile: it is consumed by the first split and read
        // again here (an L2 hit: this wave fetched it a few microseconds ago) for the layer-0 weight gradient and the
        // S1/S2 sums -- 16 registers less during the phases where dy, z and the hidden activations are live.
        constexpr bool RELOAD = (CA == 32) && X3_RELOAD_X;
        if constexpr (RELOAD) load_raw<SA>(xa, va, c, h);
        {
            X3 D;
            split16m(D, dpre, negI);
            acc = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_W1T, D, lane);
        }
        dWh[0] = wgrad_x3(SD1, SH1, dWh[0], db[1], lane, negI);
        {
            float hsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) hsv[r] = SH1[ch_of(r, h) * TLD + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) dpre[r] = hsv[r] > 0.f ? acc[r] : 0.f;
        }
        // ---- layer 0 ----
        if constexpr (VW0) {
            float ya[1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);          // channel h of this lane's pixel
            const float other = __shfl_xor(ya[0], 32);
            const float x0 = h ? other : ya[0], x1 = h ? ya[0] : other;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                w0v[2 * r] = fmaf(dpre[r], x0, w0v[2 * r]);
                w0v[2 * r + 1] = fmaf(dpre[r], x1, w0v[2 * r + 1]);
                b0v[r] += dpre[r];
            }
        } else {
            stage16(SD0, dpre, j, h);
            float ya[SA > 0 ? SA : 1];
            norm_slab<SA>(ya, xa, recA, normA, c_valid, h);
            if constexpr (CA < 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) SX[ch_of(r, h) * TLD + j] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < SA; ++s) SX[slab_ch<SA>(s, h) * TLD + j] = ya[s];
        }
        X3 D0;
        if (A.dxa != nullptr || (CB > 0 && A.dxb != nullptr)) split16m(D0, dpre, negI);
        // the emitting variant still needs xa for (z - mean); everyone else prefetches the next tile's slab a now
        float nxa[(CA == 32 && CB == 0) ? 16 : 1];
        {
            const TileCtx cn = decode_tile(tnext, tnext < T1, tpg, A.N, P, j);
            if constexpr (CA == 32 && CB == 0) load_raw<SA>(nxa, va, cn, h);
            else load_slab<SA, PK>(xa, va, ps, cn, h);
        }
        f32x16 dxa_acc = zero, dxb_acc = zero;
        if constexpr (CA == 32) {
            if (A.dxa) dxa_acc = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_WT0A, D0, lane);
        }
        if constexpr (CB == 32) {
            if (A.dxb) dxb_acc = gemm_x3<2, X3_BWD_TERMS>(zero, wl, L::PD, L::OFF_WT0B, D0, lane);
        }
        if constexpr (!VW0) {
            X3 DT;
            load_wgrad_operand<true>(DT, SD0, db[0], lane, negI);
            {
                X3 XT;
                float dummy = 0.f;
                load_wgrad_operand<false>(XT, SX, dummy, lane, negI);
                dW0a = gemm_x3_rr<X3_BWD_TERMS>(dW0a, DT, XT);
            }
            if constexpr (CB > 0) {
                // x_b takes x_a's slot (its readers were issued above)
                float yb[SB > 0 ? SB : 1];
                norm_slab<SB>(yb, xb, recB, normB, c_valid, h);
                if constexpr (CB < 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) SX[ch_of(r, h) * TLD + j] = 0.f;
                }
#pragma unroll
                for (int s = 0; s < SB; ++s) SX[slab_ch<SB>(s, h) * TLD + j] = yb[s];
                X3 XT;
                float dummy = 0.f;
                load_wgrad_operand<false>(XT, SX, dummy, lane, negI);
                dW0b = gemm_x3_rr<X3_BWD_TERMS>(dW0b, DT, XT);
                const TileCtx cn = decode_tile(tnext, tnext < T1, tpg, A.N, P, j);
                load_slab<SB, PK>(xb, vb, ps, cn, h);
            }
        }
        // ---- dx ----
        if constexpr (CA == 32) {
            if (A.dxa) {
                const int voff = lane_off<4>(vdxa, c, h);
                const int s0 = c.g * vdxa.gs4;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = dxa_acc[r];
                if constexpr (EARLY_RMW) {
                    if (rmw) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] += old[r];
                    }
                } else {
                    if (A.accumulate_a) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] += buf_load(vdxa, voff, s0 + ((r & 3) + 8 * (r >> 2)) * vdxa.ld4);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) buf_store(v[r], vdxa, voff, s0 + ((r & 3) + 8 * (r >> 2)) * vdxa.ld4);
                if constexpr (CB == 0) {
                    if (emit) {
                        // GraphNorm-backward sums of the producer of slab a over this tile: S1 = sum v, S2 = sum v (z_a - mean_a)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ch = ch_of(r, h);
                            const float mean = reinterpret_cast<const float4 *>(recA)[ch].x;
                            S1[ch * TLD + j] = c_valid ? v[r] : 0.f;
                            S2[ch * TLD + j] = xa[r] - mean;
                        }
                        const float4 *vp = reinterpret_cast<const float4 *>(S1 + j * TLD + 16 * h);
                        const float4 *up = reinterpret_cast<const float4 *>(S2 + j * TLD + 16 * h);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float4 a = vp[k], b = up[k];
                            s1 += (a.x + a.y) + (a.z + a.w);
                            s2 += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
                        }
                        s1 += __shfl_xor(s1, 32);
                        s2 += __shfl_xor(s2, 32);
                        if (h == 0) {
                            float2 o;
                            o.x = s1;
                            o.y = s2;
                            reinterpret_cast<float2 *>(A.s12part)[((long long)c.g * tpg + c.tt) * FGNN_H + j] = o;
                        }
                    }
                }
            }
        }
        if constexpr (CB == 32) {
            if (A.dxb) {
                const int voff = lane_off<4>(vdxb, c, h);
                const int s0 = c.g * vdxb.gs4;
                float vb2[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) vb2[r] = dxb_acc[r];
                if (A.accumulate_b) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) vb2[r] += buf_load(vdxb, voff, s0 + ((r & 3) + 8 * (r >> 2)) * vdxb.ld4);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) buf_store(vb2[r], vdxb, voff, s0 + ((r & 3) + 8 * (r >> 2)) * vdxb.ld4);
            }
        }
        if constexpr (CA == 32 && CB == 0) {
#pragma unroll
            for (int s = 0; s < SA; ++s) xa[s] = nxa[s];
        }
    }

    // ---- workgroup reduction of the parameter gradients (fixed order over the waves) ----
    // layout: [W0 (32*CIN) | b0 (32) | W1 (1024) | b1 (32) | W2 (1024) | b2 (32)]
    constexpr int PCOUNT = L::PCOUNT;
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] += __shfl_xor(db[l], 32);
    if constexpr (VW0) {
#pragma unroll
        for (int r = 0; r < 32; ++r) w0v[r] = half_sum(w0v[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) b0v[r] = half_sum(b0v[r]);
    }
    __syncthreads();                       // everyone done with the operand image and the tile buffers
    {
        float *red = smem + wv * PCOUNT;   // the whole LDS allocation is free now
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ch_of(r, h);
            if constexpr (VW0) {
                if (j == 0) {
                    red[o * CIN] = w0v[2 * r];
                    red[o * CIN + 1] = w0v[2 * r + 1];
                    red[32 * CIN + o] = b0v[r];
                }
            } else {
                if (j < CA) red[o * CIN + j] = dW0a[r];
                if (CB > 0 && j < CB) red[o * CIN + CA + j] = dW0b[r];
            }
        }
        int off = 32 * CIN;
#pragma unroll
        for (int l = 0; l < DEPTH; ++l) {
            if (l > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[off + ch_of(r, h) * 32 + j] = dWh[l - 1][r];
                off += 1024;
            }
            if (h == 0 && !(VW0 && l == 0)) red[off + j] = db[l];
            off += 32;
        }
    }
    __syncthreads();
    static_assert(PCOUNT % 4 == 0, "partials are summed four at a time");
    float4 *out = reinterpret_cast<float4 *>(A.wpart + (long long)blockIdx.x * PCOUNT);
    const float4 *part4 = reinterpret_cast<const float4 *>(smem);
    for (int e = threadIdx.x; e < PCOUNT / 4; e += 64 * NW) {
        float4 a = part4[e];
#pragma unroll
        for (int w = 1; w < NW; ++w) {                                  // fixed order
            const float4 b = part4[w * (PCOUNT / 4) + e];
            a.x += b.x;
            a.y += b.y;
            a.z += b.z;
            a.w += b.w;
        }
        out[e] = a;
    }
}

template <int CA, int CB, bool PK, int NWT = 8>
int launch_bwd_x3(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    constexpr int LDS = BwdX3Layout<CA, CB, NWT>::LDS_F * 4;
    constexpr int NW = NWT;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd_x3_kernel<CA, CB, PK, NWT>, LDS);
    hipLaunchKernelGGL((mlp_bwd_x3_kernel<CA, CB, PK, NWT>), dim3(a->cu_share == 2 ? BWD_WG / 2 : BWD_WG), dim3(64 * NW), LDS, st, *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int fgnn_mlp_bwd_x3(const fgnn_mlp_bwd_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_bwd_x3: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_bwd_x3: bad G=%d N=%d", a->G, a->N);
    FGNN_CHECK(fgnn_mlp_x3_supported(a->a.C, a->b.C, a->depth, 1),
               "fgnn_mlp_bwd_x3: built for depth 3 and 2, 32, 32+2, 32+32 input channels (got depth %d, %d + %d); use fgnn_mlp_bwd",
               a->depth, a->a.C, a->b.C);
    FGNN_CHECK(a->packed, "fgnn_mlp_bwd_x3: needs the operand image of fgnn_pack_x3_operands (kind 1)");
    FGNN_CHECK(!a->ranges, "fgnn_mlp_bwd_x3: no padding-tile skipping (ranges); use fgnn_mlp_bwd for ragged batches");
    FGNN_CHECK(BWD_WG == fgnn_mlp_bwd_num_workgroups(), "fgnn_mlp_bwd_x3: workgroup count differs from fgnn_mlp_bwd");
    const bool pk_a = a->xbits && a->a.C == 2, pk_b = a->xbits && a->b.C == 2;
    FGNN_CHECK((a->a.ptr || pk_a) && (a->b.C == 0 || a->b.ptr || pk_b), "fgnn_mlp_bwd_x3: slab pointer missing");
    FGNN_CHECK(!a->xbits || a->xdeg, "fgnn_mlp_bwd_x3: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK(!(a->a.C != 32 && a->dxa) && !(a->b.C != 32 && a->b.C != 0 && a->dxb),
               "fgnn_mlp_bwd_x3: input gradients exist for 32-channel slabs only; use fgnn_mlp_bwd");
    FGNN_CHECK(a->dy && a->z && a->wpart, "fgnn_mlp_bwd_x3: missing dy/z/wpart");
    FGNN_CHECK(a->coef || (a->s12 && a->znrm) || (a->s12tiles && a->znrm), "fgnn_mlp_bwd_x3: need coef, or s12 + znrm, or s12tiles + znrm");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim &&
                   G * a->dxa_gstride < lim && G * a->dxb_gstride < lim,
                   "fgnn_mlp_bwd_x3: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd_x3: too many tiles");
    FGNN_CHECK(!a->s12tiles || a->b.C > 0, "fgnn_mlp_bwd_x3: s12tiles is built into the two-slab kernels only");
    FGNN_CHECK(!a->s12tiles || fgnn_mlp_bwd_coef_tiles_supported(a->G, a->N),
               "fgnn_mlp_bwd_x3: s12tiles needs a workgroup's tile range to span <= %d graphs (G=%d N=%d)", FGNN_BWD_COEF_GRAPHS, a->G, a->N);
    hipStream_t st = (hipStream_t)stream;
    const int ca = a->a.C, cb = a->b.C;
    if (a->xbits) {
        if (ca == 2 && cb == 0) return launch_bwd_x3<2, 0, true>(a, tpg, (int)total, st);
        if (ca == 32 && cb == 2) return launch_bwd_x3<32, 2, true>(a, tpg, (int)total, st);
        fgnn_set_error("fgnn_mlp_bwd_x3: xbits needs a 2-channel slab (2 or 32+2 input channels)");
        return 1;
    }
    if (ca == 2 && cb == 0) return launch_bwd_x3<2, 0, false>(a, tpg, (int)total, st);
    if (ca == 32 && cb == 0) return launch_bwd_x3<32, 0, false>(a, tpg, (int)total, st);
    if (ca == 32 && cb == 2) return launch_bwd_x3<32, 2, false>(a, tpg, (int)total, st);
    return launch_bwd_x3<32, 32, false>(a, tpg, (int)total, st);
}
