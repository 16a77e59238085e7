// MlpBlock_Real backward of a TWO-slab MLP (mlp3 of an FGNN block: input [mult ; in], models/blocks_emb.py:29-36, the autograd of
// models/layers.py:126-131 plus the GraphNorm backward of :68-80 folded into dz) on 16-pixel tiles / v_mfma_f32_16x16x4_f32
// (round 6; the 32-pixel original: mlp_bwd.hip, same arguments, partial rows and results to fp32 rounding).
//
// What changes against mlp_bwd.hip:
//   * the unit of work is a 16-pixel HALF tile (fgnn_t16.h), assigned to the eight waves of a workgroup round-robin: 39.5 halves per
//     workgroup at the benchmarked shape = 4.94 per wave (five rounds, the last one 94 % full) instead of 2.47 32-pixel tiles per
//     wave (three rounds, the last one half empty with one wave per SIMD);
//   * per-graph records (GraphNorm record of the input slab, dz coefficients) live in registers for all halves of a graph, the hidden
//     activations stay in registers for the ReLU masks, the input is normalised once, the padding mask is applied to dz only;
//   * every global load of a half is requested while the previous half computes, in straight-line code (so that the compiler's
//     s_waitcnt for the one that is needed never drains the others or the stores), the operand image arrives by global_load_lds.
// The recomputed hidden activations follow the forward's fma sequence bit for bit (fgnn_t16.h).
// Built for depth 3, slab a = 32 raw channels (mult), slab b = 32 channels (a block's input, normalised on load) or 2 raw channels
// (block 1: the model input, dense or bit-packed); dx for both slabs (stored, not accumulated) or for slab a only.  Anything else:
// fgnn_mlp_bwd.
#include "fgnn_t16.h"
#include "fgnn_pack.h"

namespace {

using namespace t16;

constexpr int BWD_WG = 256;          // persistent workgroups (one per CU) = rows of wpart
constexpr int NW = 8;                // waves per workgroup (2 per SIMD)

DEVI float4 coef_from_sums(const float4 n, const float2 sv, float nv) {
    const float m = nv * nv;
    float4 k;
    k.x = n.x;
    k.y = n.y;
    k.z = m > 0.f ? -n.y * sv.y * n.w / m : 0.f;
    k.w = m > 0.f ? -n.y * sv.x / m : 0.f;
    return k;
}
// vertex count of graph g through a buffer descriptor (see mlp_bwd_pair_t16.hip: a select of two addresses would be a FLAT load)
DEVI int graph_nv(const rsrc_t &rnv, bool ragged, int g, int N) {
    const int v = __builtin_amdgcn_raw_buffer_load_b32(rnv, g * 4, 0, 0);
    return __builtin_amdgcn_readfirstlane(ragged ? v : N);
}

template <int CB>
struct Layout16 {
    static constexpr int DEPTH = 3;
    static constexpr PkBwd PK = pk_bwd(32, CB, DEPTH);                    // fgnn_pack.h, image kind 5
    static constexpr int OFF_W0A = PK.off_w1a, OFF_W0B = PK.off_w1b, OFF_W1 = PK.off_wh, OFF_WT1 = PK.off_wt, OFF_WT2 = PK.off_wt + 16;
    static constexpr int OFF_WT0A = PK.off_wt0a, OFF_WT0B = PK.off_wt0b;
    static constexpr int BIAS_F = PK.bias_f;
    static constexpr int WEIGHT_F = pk_pad_floats(PK.floats);
    static constexpr int NSLOT = 5;                                       // per wave: x_a, x_b, h1, h2 / dpre_1, dz / dpre_0
    static constexpr int CIN = 32 + CB;
    static constexpr int PCOUNT = 32 * CIN + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int TILE_OFF = WEIGHT_F;
    static constexpr int REC_OFF = TILE_OFF + NW * NSLOT * TILE_F;        // per wave: {coef[32], nrm_b[32]} float4 (graph changes only)
    static constexpr int WGK_OFF = REC_OFF + NW * 256;                    // workgroup cache of dz coefficient records (s12tiles)
    static constexpr int LIVE_OFF = WGK_OFF + FGNN_BWD_COEF_GRAPHS * 128;  // SKIP: the range's live tiles (build_live_list) + NW counters
    static constexpr int MAIN_F = LIVE_OFF + LIVE_LIST_CAP + NW;
    static constexpr int RED_F = NW * PCOUNT;
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

// SKIP (ragged batches with ranges): the workgroup's tile range comes from fgnn_ragged_tile_ranges, 32-pixel tiles without a valid
// pixel are stepped over (both halves of a live tile are processed: every pixel of a live tile is written, as the consumers expect).
// CB = 32: slab b normalised on load (NB).  CB = 2: raw 2-channel slab, PKD = expanded from the bit-packed adjacency.
template <int CB, bool PKD, bool SKIP, bool DXB>
__global__ __launch_bounds__(64 * NW, 2) void mlp_bwd_t16_kernel(const fgnn_mlp_bwd_args A, const int tpg, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Layout16<CB>;
    constexpr bool NB = CB == 32;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    young_prio(3, wv, NW);
    const int px = lane & 15, q = lane >> 4;
    const int P2 = A.N * A.N, hpg = 2 * tpg;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vb = make_view(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    PackedSrc ps = {};
    if constexpr (PKD) ps = make_packed_src(A.xbits, A.xdeg, A.G, A.N);
    const View vdy = make_view(A.dy, A.dgstride, A.ldd, A.G);
    const View vz = make_view(A.z, A.zgstride, A.ldz, A.G);
    const View vdxa = make_view(A.dxa, A.dxa_gstride, A.dxa_ld, A.G);
    const View vdxb = make_view(A.dxb, A.dxb_gstride, A.dxb_ld, A.G);

    float *wl = smem;
    float *my = smem + L::TILE_OFF + wv * (L::NSLOT * TILE_F);
    float *XA = my, *XB = my + TILE_F, *S0 = my + 2 * TILE_F, *S1 = my + 3 * TILE_F, *S2 = my + 4 * TILE_F;
    const int lane_base = tile_lane_base(px, q);

    f32x4 dW0a[4], dW0b[4], dW1[4], dW2[4];
    float db0[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f}, db2[2] = {0.f, 0.f}, dbx[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) dW0a[k] = dW0b[k] = dW1[k] = dW2[k] = zero4();

    const int nwg = gridDim.x;
    const int qq = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * qq + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + qq + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }
    const int H0 = 2 * T0, H1 = 2 * T1;                 // this workgroup's halves
    const bool ragged = A.nvalid != nullptr;
    const rsrc_t rnv = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(A.nvalid), 0, ragged ? A.G * 4 : 0, 0x00020000);
    const float rcpN = 1.f / (float)A.N;
    const bool from_tiles = A.s12tiles != nullptr;
    const int g0 = T0 / tpg;
    float4 *wgK = reinterpret_cast<float4 *>(smem + L::WGK_OFF);

    // the wave's next half at or after h (h, H1: wave-uniform).  Dense batches: every NW-th half.  SKIP: the waves take the LIVE halves of
    // the range in turn (live_cnt = live halves seen so far; next_owned_live_tile_p in fgnn_common.h says why), so every wave scans the
    // whole range and counts
    int live_cnt = 0;
    int *live_list = reinterpret_cast<int *>(smem + L::LIVE_OFF);
    int nlive = 0, lk = wv - NW;                    // (SKIP) live tiles of the range; this wave's index among their halves
    bool use_list = false;
    if constexpr (SKIP) {                            // (before anything is in flight: its barriers would drain it)
        nlive = build_live_list(live_list, live_list + LIVE_LIST_CAP, T0, T1, tpg, FGNN_TILE, A.N, A.nvalid, threadIdx.x, 64 * NW);
        use_list = nlive <= LIVE_LIST_CAP;
    }
    auto next_half = [&](int h) {
        if constexpr (SKIP) {
            if (use_list) {                         // entry k = half (k & 1) of live tile k >> 1; halves past the plane's end are nobody's
                int r = H1;
                while (true) {
                    lk += NW;
                    if ((lk >> 1) >= nlive) break;
                    const int t = live_list[lk >> 1], g = t / tpg, hh = 2 * (t - g * tpg) + (lk & 1);
                    if (hh * 16 < P2) {
                        r = 2 * t + (lk & 1);
                        break;
                    }
                }
                return __builtin_amdgcn_readfirstlane(r);
            }
            while (h < H1) {
                const int g = h / hpg, hh = h - g * hpg;
                const int nv = A.nvalid[g];
                if (hh * 16 < P2 && tile_live(hh >> 1, A.N, nv)) {
                    const bool mine = live_cnt % NW == wv;
                    ++live_cnt;
                    if (mine) break;
                    ++h;
                } else if ((hh >> 1) * FGNN_TILE / A.N >= nv) {
                    h = (g + 1) * hpg;             // rows >= nv are padding up to the end of the graph
                } else {
                    ++h;
                }
            }
            return __builtin_amdgcn_readfirstlane(h < H1 ? h : H1);
        } else {
            while (h < H1) {
                const int g = h / hpg, hh = h - g * hpg;
                if (hh * 16 < P2) break;
                h += NW;
            }
            return __builtin_amdgcn_readfirstlane(h);
        }
    };

    // ---- prologue: the operand image (straight into LDS), the first half's x ----
    pk_glds<NW>(smem, A.packed, L::WEIGHT_F, wv, lane);

    float mean[8], av[8], beta[8], kx[8], ky[8], kz[8], kw[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        mean[s] = 0.f;
        av[s] = 1.f;
        beta[s] = (NB && A.b.nrm && A.b.beta) ? A.b.beta[chan_s(s) + chan_q(q)] : 0.f;
        kx[s] = ky[s] = kz[s] = kw[s] = 0.f;
    }
    float4 *rec = reinterpret_cast<float4 *>(smem + L::REC_OFF) + wv * 64;
    auto fetch_records = [&](int g, int nv) {       // one channel per lane (lanes 0..31); dz coefficients from the workgroup cache with s12tiles
        if (lane < 32) {
            float4 k4;
            if (from_tiles) k4 = wgK[(g - g0) * 32 + lane];
            else if (A.coef) k4 = reinterpret_cast<const float4 *>(A.coef)[(long long)g * FGNN_H + lane];
            else {
                const float4 n = reinterpret_cast<const float4 *>(A.znrm)[(long long)g * FGNN_H + lane];
                const float2 sv = reinterpret_cast<const float2 *>(A.s12)[(long long)g * FGNN_H + lane];
                k4 = coef_from_sums(n, sv, (float)nv);
            }
            rec[lane] = k4;
            if (NB && A.b.nrm) rec[32 + lane] = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)g * A.b.C + lane];
        }
    };
    auto read_records = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 k4 = rec[chan_s(s) + chan_q(q)];
            kx[s] = k4.x;
            ky[s] = k4.y;
            kz[s] = k4.z;
            kw[s] = k4.w;
            if (NB && A.b.nrm) {
                const float4 n = rec[32 + chan_s(s) + chan_q(q)];
                mean[s] = n.x;
                av[s] = n.y;
            }
        }
    };

    // slab loads of half h: xa[8] (32 raw channels), xb (CB = 32: 8 registers; CB = 2: one -- channel q for q < 2)
    constexpr int NXB = CB == 32 ? 8 : 1;
    float xa[8], xb[NXB], dyr[8], zr[8];
    auto load_x = [&](int h) {
        const bool act = h < H1;
        const int g = __builtin_amdgcn_readfirstlane(act ? h / hpg : 0);
        const int p = (act ? h - g * hpg : 0) * 16 + px;
        const bool inb = act && p < P2;
        load8(xa, va, lane_voff(va, q, p, inb), g * va.gs4);
        if constexpr (CB == 32) {
            load8(xb, vb, lane_voff(vb, q, p, inb), g * vb.gs4);
        } else if constexpr (PKD) {
            int i, jj;
            row_col(p, A.N, rcpN, i, jj);
            const int ob = (inb && q == 0) ? (i * ps.words + (jj >> 5)) * 4 : OOB_OFF;
            const int od = (inb && q == 1 && i == jj) ? i * 4 : OOB_OFF;
            const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ps.bits, ob, g * ps.N * ps.words * 4, 0);
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ps.deg, od, g * ps.N * 4, 0));
            xb[0] = q == 0 ? (((w >> (jj & 31)) & 1u) ? 1.f : 0.f) : d;       // (needs w: resolved at the use of xb)
        } else {
            xb[0] = buf_load(vb, (inb && q < 2) ? q * vb.ld4 + 4 * p : OOB_OFF, g * vb.gs4);
        }
    };
    auto load_dyz = [&](int h) {
        const bool act = h < H1;
        const int g = __builtin_amdgcn_readfirstlane(act ? h / hpg : 0);
        const int p = (act ? h - g * hpg : 0) * 16 + px;
        const bool inb = act && p < P2;
        load8(dyr, vdy, lane_voff(vdy, q, p, inb), g * vdy.gs4);
        load8(zr, vz, lane_voff(vz, q, p, inb), g * vz.gs4);
    };

    int h = next_half(SKIP ? H0 : H0 + wv);
    load_x(h);
    int cached_g = -1, cur_nv = A.N;

    if (from_tiles) {
        // dz coefficients of the (few) graphs this workgroup touches from the per-tile S1/S2 sums its consumer left behind
        // (the work of fgnn_gn_bwd_coef_tiles without its launch; mlp_bwd.hip)
        const int g1 = T1 > T0 ? (T1 - 1) / tpg : g0 - 1;
        float2 *scr = reinterpret_cast<float2 *>(smem + L::TILE_OFF);     // [16 slices][32 channels]; the tiles are free here
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
                wgK[(g - g0) * 32 + cc] = coef_from_sums(n, sv, (float)graph_nv(rnv, ragged, g, A.N));
            }
            __syncthreads();
        }
    }
    if (h < H1) {
        const int g = h / hpg;
        cur_nv = graph_nv(rnv, ragged, g, A.N);
        fetch_records(g, cur_nv);
        cached_g = g;
    }
    if constexpr (CB == 2) {                         // the slab-b tile holds two rows (channels 0, 1 = rows 0, 8); the rest of its first block is read as zeros
        for (int e = lane; e < 16 * TLD; e += 64) XB[e] = 0.f;          // (rows 0 and 8 are rewritten per half)
    }
    __syncthreads();
    load_dyz(h);                                     // (after the barrier, which drains every load: see mlp_bwd_pair_t16.hip)
    if (cached_g >= 0) read_records();

#ifdef FGNN_PRIOB          // measurement switch: static priority for the younger waves (1) / the older waves (2)
    if ((FGNN_PRIOB == 1) == (wv >= 4)) __builtin_amdgcn_s_setprio(1);
#endif
    while (h < H1) {
        const int hn = next_half(SKIP ? h + 1 : h + NW);
        const int g = __builtin_amdgcn_readfirstlane(h / hpg), hh = h - g * hpg;
        const int p = hh * 16 + px;
        const bool inb = p < P2;
        if (g != cached_g) {
            cur_nv = graph_nv(rnv, ragged, g, A.N);
            fetch_records(g, cur_nv);
            read_records();
            cached_g = g;
        }
        bool valid = inb;
        if (ragged) {
            int i, jj;
            row_col(p, A.N, rcpN, i, jj);
            valid = inb && i < cur_nv && jj < cur_nv;
        }
        const bool full = __ballot(valid) == ~0ull;

        // ---- forward recompute of the hidden activations (bit-identical to the forward's chain) ----
        float h1[8], h2[8];
        {
            float yb[NXB];
            if constexpr (NB) {
#pragma unroll
                for (int s = 0; s < 8; ++s) yb[s] = (xb[s] - mean[s]) * av[s] + beta[s];
            } else {
                yb[0] = xb[0];
            }
            stage8(XA, lane_base, xa);
            if constexpr (CB == 32) {
                stage8(XB, lane_base, yb);
            } else {                                   // rows 0, 1 of the slab-b tile; rows 2..15 of its first block were zeroed once
                if (q < 2) XB[(FGNN_ROWMAP ? 8 : 1) * q * TLD + px] = yb[0];     // channel q = chan(0, 2 q): row 4 * (2 q)
            }
            f32x4 acc[2];
            load_bias(acc, wl + L::BIAS_F, 0, q);
            gemm32<L::OFF_W0A>(acc, wl, xa, lane);
            if constexpr (CB == 32) gemm32<L::OFF_W0B>(acc, wl, yb, lane);
            else gemm2<L::OFF_W0B>(acc, wl, q < 2 ? yb[0] : 0.f, lane);
            load_x(hn);                               // the next half's input slabs into the registers just consumed
#pragma unroll
            for (int s = 0; s < 8; ++s) h1[s] = relu1(acc[s >> 2][s & 3]);
            stage8(S0, lane_base, h1);
            load_bias(acc, wl + L::BIAS_F, 1, q);
            gemm32<L::OFF_W1>(acc, wl, h1, lane);
#pragma unroll
            for (int s = 0; s < 8; ++s) h2[s] = relu1(acc[s >> 2][s & 3]);
            stage8(S1, lane_base, h2);
        }
        // ---- dz from (dy, z, coef); the ONLY place the padding mask is applied ----
        float dpre[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) dpre[s] = fmaf(kz[s], zr[s] - kx[s], fmaf(ky[s], dyr[s], kw[s]));
        if (!full) {
#pragma unroll
            for (int s = 0; s < 8; ++s) dpre[s] = valid ? dpre[s] : 0.f;
        }
        stage8(S2, lane_base, dpre);
        load_dyz(hn);                                 // the next half's dy / z into the registers just consumed
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 2 ----
        {
            f32x4 a2[2];
            a2[0] = a2[1] = zero4();
            gemm32<L::OFF_WT2>(a2, wl, dpre, lane);
            wgrad16(dW2, db2, S2, S1, lane);
#pragma unroll
            for (int s = 0; s < 8; ++s) dpre[s] = h2[s] > 0.f ? a2[s >> 2][s & 3] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        stage8(S1, lane_base, dpre);                  // h2's tile is dead (LDS is in order within a wave)
        // ---- layer 1 ----
        {
            f32x4 a2[2];
            a2[0] = a2[1] = zero4();
            gemm32<L::OFF_WT1>(a2, wl, dpre, lane);
            wgrad16(dW1, db1, S1, S0, lane);
#pragma unroll
            for (int s = 0; s < 8; ++s) dpre[s] = h1[s] > 0.f ? a2[s >> 2][s & 3] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        stage8(S2, lane_base, dpre);                  // dz's tile is dead
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 0: input gradients of both slabs, weight gradients against both slab tiles ----
        {
            f32x4 dx[2];
            float v[8];
            dx[0] = dx[1] = zero4();
            gemm32<L::OFF_WT0A>(dx, wl, dpre, lane);
            wgrad16(dW0a, db0, S2, XA, lane);
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = dx[s >> 2][s & 3];
            store8(v, vdxa, lane_voff(vdxa, q, p, inb), g * vdxa.gs4);
            if constexpr (DXB) {
                dx[0] = dx[1] = zero4();
                gemm32<L::OFF_WT0B>(dx, wl, dpre, lane);
            }
            if constexpr (CB == 32) {
                wgrad16(dW0b, dbx, S2, XB, lane);
            } else {                                   // 2-channel slab: only the first 16-column block of the tile holds anything
                const int i = lane & 15;
                const float4 a0 = *reinterpret_cast<const float4 *>(S2 + i * TLD + 4 * q);
                const float4 a1 = *reinterpret_cast<const float4 *>(S2 + (16 + i) * TLD + 4 * q);
                const float4 b0 = *reinterpret_cast<const float4 *>(XB + i * TLD + 4 * q);
                dW0b[0] = mfma16(a0.x, b0.x, dW0b[0]);
                dW0b[2] = mfma16(a1.x, b0.x, dW0b[2]);
                dW0b[0] = mfma16(a0.y, b0.y, dW0b[0]);
                dW0b[2] = mfma16(a1.y, b0.y, dW0b[2]);
                dW0b[0] = mfma16(a0.z, b0.z, dW0b[0]);
                dW0b[2] = mfma16(a1.z, b0.z, dW0b[2]);
                dW0b[0] = mfma16(a0.w, b0.w, dW0b[0]);
                dW0b[2] = mfma16(a1.w, b0.w, dW0b[2]);
            }
            if constexpr (DXB) {
#pragma unroll
                for (int s = 0; s < 8; ++s) v[s] = dx[s >> 2][s & 3];
                store8(v, vdxb, lane_voff(vdxb, q, p, inb), g * vdxb.gs4);
            }
        }
        h = hn;
    }

    // ---- workgroup reduction of the parameter gradients (fixed order over the waves) ----
    // layout: [W0 (32*CIN) | b0 (32) | W1 (1024) | b1 (32) | W2 (1024) | b2 (32)]
    constexpr int PCOUNT = L::PCOUNT, CIN = L::CIN;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        db0[b] += __shfl_xor(db0[b], 16);
        db0[b] += __shfl_xor(db0[b], 32);
        db1[b] += __shfl_xor(db1[b], 16);
        db1[b] += __shfl_xor(db1[b], 32);
        db2[b] += __shfl_xor(db2[b], 16);
        db2[b] += __shfl_xor(db2[b], 32);
    }
    (void)dbx;
    __syncthreads();                       // everyone done with the operand image and the tile buffers
    {
        float *red = smem + wv * PCOUNT;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = row_chan(16 * mb + 4 * q + r), c = row_chan(16 * nb + px);
                    red[o * CIN + c] = dW0a[2 * mb + nb][r];
                    if (c < CB) red[o * CIN + 32 + c] = dW0b[2 * mb + nb][r];
                    red[32 * CIN + 32 + o * 32 + c] = dW1[2 * mb + nb][r];
                    red[32 * CIN + 32 + 1056 + o * 32 + c] = dW2[2 * mb + nb][r];
                }
        if (q == 0) {
            const int c0 = row_chan(px), c1 = row_chan(16 + px);
            red[32 * CIN + c0] = db0[0];
            red[32 * CIN + c1] = db0[1];
            red[32 * CIN + 32 + 1024 + c0] = db1[0];
            red[32 * CIN + 32 + 1024 + c1] = db1[1];
            red[32 * CIN + 32 + 1056 + 1024 + c0] = db2[0];
            red[32 * CIN + 32 + 1056 + 1024 + c1] = db2[1];
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

template <int CB, bool PKD, bool SKIP, bool DXB>
int launch16(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    constexpr int LDS = Layout16<CB>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd_t16_kernel<CB, PKD, SKIP, DXB>, LDS);
    hipLaunchKernelGGL((mlp_bwd_t16_kernel<CB, PKD, SKIP, DXB>), dim3(a->cu_share == 2 && !SKIP ? BWD_WG / 2 : BWD_WG), dim3(64 * NW), LDS, st, *a,
                       tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CB, bool PKD, bool DXB>
int launch16s(const fgnn_mlp_bwd_args *a, int tpg, int total, hipStream_t st) {
    return a->ranges ? launch16<CB, PKD, true, DXB>(a, tpg, total, st) : launch16<CB, PKD, false, DXB>(a, tpg, total, st);
}

}  // namespace

// 1 when fgnn_mlp_bwd_t16 is built for this argument block (else: fgnn_mlp_bwd)
extern "C" int fgnn_mlp_bwd_t16_supported(const fgnn_mlp_bwd_args *a) {
    if (!a || a->depth != 3 || a->a.C != 32 || a->a.nrm || !a->dxa || a->accumulate_a || a->accumulate_b || !a->packed || a->N > 256) return 0;
    if (a->b.C == 32) return (a->b.nrm && a->dxb && !a->xbits) ? 1 : 0;
    if (a->b.C == 2) return (!a->b.nrm && !a->dxb) ? 1 : 0;
    return 0;
}

// Same contract as fgnn_mlp_bwd for the argument blocks fgnn_mlp_bwd_t16_supported accepts; `packed` is an image of kind 5.
extern "C" int fgnn_mlp_bwd_t16(const fgnn_mlp_bwd_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_bwd_t16: null args");
    FGNN_CHECK(fgnn_mlp_bwd_t16_supported(a), "fgnn_mlp_bwd_t16: built for depth 3, slab a = 32 raw channels with dx stored, slab b = 32 normalised "
               "channels with dx stored or 2 raw channels without dx, an operand image of kind 5, N <= 256; use fgnn_mlp_bwd");
    FGNN_CHECK(a->G > 0 && a->N > 0 && a->a.ptr && (a->b.ptr || (a->xbits && a->b.C == 2)), "fgnn_mlp_bwd_t16: bad G / N / slabs");
    FGNN_CHECK(!a->xbits || a->xdeg, "fgnn_mlp_bwd_t16: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK(a->dy && a->z && a->wpart, "fgnn_mlp_bwd_t16: missing dy/z/wpart");
    FGNN_CHECK(a->coef || (a->s12 && a->znrm) || (a->s12tiles && a->znrm), "fgnn_mlp_bwd_t16: need coef, or s12 + znrm, or s12tiles + znrm");
    FGNN_CHECK(BWD_WG == fgnn_mlp_bwd_num_workgroups() && BWD_WG == FGNN_RANGE_WG, "fgnn_mlp_bwd_t16: workgroup count differs from fgnn_mlp_bwd");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim &&
                   G * a->dxa_gstride < lim && G * a->dxb_gstride < lim,
                   "fgnn_mlp_bwd_t16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    for (int l = 0; l < a->depth; ++l) FGNN_CHECK(a->W[l] && a->bias[l], "fgnn_mlp_bwd_t16: missing weights layer %d", l);
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 29), "fgnn_mlp_bwd_t16: too many tiles");
    FGNN_CHECK(!a->ranges || (a->nvalid && !a->s12tiles), "fgnn_mlp_bwd_t16: ranges need nvalid and exclude s12tiles");
    FGNN_CHECK(!a->s12tiles || fgnn_mlp_bwd_coef_tiles_supported(a->G, a->N), "fgnn_mlp_bwd_t16: s12tiles needs a workgroup's tile range to span <= %d graphs",
               FGNN_BWD_COEF_GRAPHS);
    hipStream_t st = (hipStream_t)stream;
    if (a->b.C == 32) return launch16s<32, false, true>(a, tpg, (int)total, st);
    if (a->xbits) return launch16s<2, true, false>(a, tpg, (int)total, st);
    return launch16s<2, false, false>(a, tpg, (int)total, st);
}
