// Fused MlpBlock_Real forward (conv1x1 + ReLU chain, last conv linear: models/layers.py:126-131, the GraphNorm reductions of
// :72-73 as tile partials) on 16-pixel tiles / v_mfma_f32_16x16x4_f32 (round 6; the 32-pixel original: mlp_fwd.hip).
//
// Same arguments as fgnn_mlp_fwd and the same z, BIT FOR BIT: the conv chain runs through the same sequence of fused multiply-adds
// (fgnn_t16.h).  What changes:
//   * the unit of work -- and of the tile statistics -- is a 16-pixel HALF tile: `part` / `cnt` hold fgnn_mlp_fwd_t16_records(N) =
//     2 * fgnn_tiles_per_graph(N) records per graph (the consumers take that count: fgnn_gn_finalize_r, fgnn_gn_finalize2_r,
//     fgnn_chan_matmul_fwd_fin_ord_r, fgnn_colmax_fwd_fin_r).  39.5 halves per workgroup at the benchmarked shape are 2.47 per wave
//     (three rounds, the last one half full) instead of 1.23 32-pixel tiles per wave (two rounds, the second a quarter full);
//   * per-graph records of the input slabs live in registers for all halves of a graph; every load of a half is requested while the
//     previous half computes; the operand image arrives by global_load_lds.
// Depth 3; slab a = 32 channels (normalised on load or raw) or the 2-channel model input; slab b (NMLP = 1 only) = 32 channels or
// the 2-channel model input; the 2-channel slab dense or expanded from the bit-packed adjacency.  Anything else: fgnn_mlp_fwd.
#include "fgnn_t16.h"
#include "fgnn_pack.h"

namespace {

using namespace t16;

#ifndef FGNN_FWD_WAVES
#define FGNN_FWD_WAVES 16
#endif
constexpr int FWD_WAVES = FGNN_FWD_WAVES;        // 4 per SIMD (<= 128 VGPRs); one workgroup per CU

template <int CA, int CB, int NMLP>
struct FwdLayout16 {
    static constexpr int NW = FWD_WAVES;
    static constexpr PkFwd PK = pk_fwd(CA, CB, 3);                        // fgnn_pack.h, image kind 4
    static constexpr int OFF_W0A = PK.off_w1a, OFF_W0B = PK.off_w1b, OFF_W1 = PK.off_wh, OFF_W2 = PK.off_wh + 16;
    static constexpr int BIAS_F = PK.bias_f;
    static constexpr int MLP_F = PK.floats;
    static constexpr int WEIGHT_F = pk_pad_floats(NMLP * MLP_F);
    static constexpr int TILE_OFF = WEIGHT_F;
    static constexpr int REC_OFF = TILE_OFF + NW * TILE_F;                // per wave: {nrm_a[32], nrm_b[32]} float4 (graph changes only)
    static constexpr int LDS_F = REC_OFF + NW * 256;
};

template <int CA, int CB, int NMLP, bool PKD, bool SKIP>
__global__ __launch_bounds__(64 * FWD_WAVES, (FWD_WAVES + 3) / 4) void mlp_fwd_t16_kernel(const fgnn_mlp_fwd_args A, const int tpg, const int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = FwdLayout16<CA, CB, NMLP>;
    constexpr int NW = L::NW;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = lane & 15, q = lane >> 4;
    const int P2 = A.N * A.N, hpg = 2 * tpg;
    const View va = make_view(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View vb = make_view(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    PackedSrc ps = {};
    if constexpr (PKD) ps = make_packed_src(A.xbits, A.xdeg, A.G, A.N);
    View vz[NMLP];
#pragma unroll
    for (int m = 0; m < NMLP; ++m) vz[m] = make_view(A.z[m], FGNN_H * A.ldz, A.ldz, A.G);

    float *wl = smem;
    float *tl = smem + L::TILE_OFF + wv * TILE_F;
    float4 *rec = reinterpret_cast<float4 *>(smem + L::REC_OFF) + wv * 64;
    const int lane_base = tile_lane_base(px, q);

    int T0, T1;
    {
        const int nwg = gridDim.x;
        const int qq = total_tiles / nwg, rem = total_tiles % nwg;
        T0 = blockIdx.x * qq + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
        T1 = T0 + qq + ((int)blockIdx.x < rem ? 1 : 0);
        if constexpr (SKIP) {
            T0 = A.ranges[blockIdx.x];
            T1 = A.ranges[blockIdx.x + 1];
        }
    }
    const int H0 = 2 * T0, H1 = 2 * T1;                 // this workgroup's halves
    const bool ragged = A.nvalid != nullptr;
    const rsrc_t rnv = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(A.nvalid), 0, ragged ? A.G * 4 : 0, 0x00020000);
    const float rcpN = 1.f / (float)A.N;
    const bool normA = CA == 32 && A.a.nrm != nullptr, normB = CB == 32 && A.b.nrm != nullptr;

    // first live half at or after h: halves past the plane and (SKIP) halves of 32-pixel tiles without a valid pixel are stepped over
    // (their records are written as empty after the main loop)
    auto next_half = [&](int h) {
        while (h < H1) {
            const int g = h / hpg, hh = h - g * hpg;
            bool live = hh * 16 < P2;
            if constexpr (SKIP) live = live && tile_live(hh >> 1, A.N, A.nvalid[g]);
            if (live) break;
            h += NW;
        }
        return __builtin_amdgcn_readfirstlane(h);
    };

    pk_glds<NW>(smem, A.packed, L::WEIGHT_F, wv, lane);

    constexpr int NXA = CA == 32 ? 8 : 1, NXB = CB == 32 ? 8 : 1;
    float ma[NXA], aa[NXA], ba[NXA], mb[NXB], ab[NXB], bb[NXB];
#pragma unroll
    for (int s = 0; s < NXA; ++s) {
        ma[s] = 0.f;
        aa[s] = 1.f;
        ba[s] = (normA && A.a.beta) ? A.a.beta[chan_s(s) + chan_q(q)] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < NXB; ++s) {
        mb[s] = 0.f;
        ab[s] = 1.f;
        bb[s] = (normB && A.b.beta) ? A.b.beta[chan_s(s) + chan_q(q)] : 0.f;
    }
    auto fetch_records = [&](int g) {
        if (lane < 32) {
            if (normA) rec[lane] = reinterpret_cast<const float4 *>(A.a.nrm)[(long long)g * A.a.C + lane];
            if (normB) rec[32 + lane] = reinterpret_cast<const float4 *>(A.b.nrm)[(long long)g * A.b.C + lane];
        }
    };
    auto read_records = [&]() {
        if constexpr (CA == 32) {
            if (normA) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float4 n = rec[chan_s(s) + chan_q(q)];
                    ma[s] = n.x;
                    aa[s] = n.y;
                }
            }
        }
        if constexpr (CB == 32) {
            if (normB) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float4 n = rec[32 + chan_s(s) + chan_q(q)];
                    mb[s] = n.x;
                    ab[s] = n.y;
                }
            }
        }
    };

    // 2-channel slab of half h: lanes q = 0, 1 carry channels 0, 1 (dense: from memory; PKD: bit / degree of the packed adjacency)
    auto load2 = [&](float &x, const View &v, int g, int p, bool inb) {
        if constexpr (PKD) {
            int i, jj;
            row_col(p, A.N, rcpN, i, jj);
            const int ob = (inb && q == 0) ? (i * ps.words + (jj >> 5)) * 4 : OOB_OFF;
            const int od = (inb && q == 1 && i == jj) ? i * 4 : OOB_OFF;
            const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ps.bits, ob, g * ps.N * ps.words * 4, 0);
            const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ps.deg, od, g * ps.N * 4, 0));
            x = q == 0 ? (((w >> (jj & 31)) & 1u) ? 1.f : 0.f) : d;
        } else {
            x = buf_load(v, (inb && q < 2) ? q * v.ld4 + 4 * p : OOB_OFF, g * v.gs4);
        }
    };
    float xa[NXA], xb[NXB];
    auto load_x = [&](int h) {
        const bool act = h < H1;
        const int g = __builtin_amdgcn_readfirstlane(act ? h / hpg : 0);
        const int p = (act ? h - g * hpg : 0) * 16 + px;
        const bool inb = act && p < P2;
        if constexpr (CA == 32) load8(xa, va, lane_voff(va, q, p, inb), g * va.gs4);
        else load2(xa[0], va, g, p, inb);
        if constexpr (CB == 32) load8(xb, vb, lane_voff(vb, q, p, inb), g * vb.gs4);
        else if constexpr (CB == 2) load2(xb[0], vb, g, p, inb);
    };

    int h = next_half(H0 + wv);
    load_x(h);
    int cached_g = -1, cur_nv = A.N;
    if (h < H1) {
        const int g = h / hpg;
        cur_nv = ragged ? __builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(rnv, g * 4, 0, 0)) : A.N;
        fetch_records(g);
        cached_g = g;
    }
    __syncthreads();
    if (cached_g >= 0) read_records();

    while (h < H1) {
        const int hn = next_half(h + NW);
        const int g = __builtin_amdgcn_readfirstlane(h / hpg), hh = h - g * hpg;
        const int p = hh * 16 + px;
        const bool inb = p < P2;
        if (g != cached_g) {
            cur_nv = ragged ? __builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(rnv, g * 4, 0, 0)) : A.N;
            fetch_records(g);
            read_records();
            cached_g = g;
        }
        bool valid = inb;
        if (ragged) {
            int i, jj;
            row_col(p, A.N, rcpN, i, jj);
            valid = inb && i < cur_nv && jj < cur_nv;
        }
        const unsigned vmask = (unsigned)__ballot(valid) & 0xffffu;      // bit px = pixel valid
        const bool full = vmask == 0xffffu;
        const float cnt = (float)__popc(vmask);
        const float inv = cnt > 0.f ? 1.f / cnt : 0.f;

        // normalise on load (the forward applies the padding mask to its input, as the 32-pixel kernel does: z of a padding pixel is
        // written as 0 whatever the chain computes there, so this is only needed for ... nothing -- left out)
        float ya[NXA], yb[NXB];
#pragma unroll
        for (int s = 0; s < NXA; ++s) ya[s] = CA == 32 ? (xa[s] - ma[s]) * aa[s] + ba[s] : xa[s];
#pragma unroll
        for (int s = 0; s < NXB; ++s) yb[s] = CB == 32 ? (xb[s] - mb[s]) * ab[s] + bb[s] : xb[s];
        load_x(hn);                                   // the next half's input slabs into the registers just consumed

#pragma unroll
        for (int m = 0; m < NMLP; ++m) {
            const float *wm = wl + m * L::MLP_F;
            f32x4 acc[2];
            load_bias(acc, wm + L::BIAS_F, 0, q);
            if constexpr (CA == 32) gemm32<L::OFF_W0A>(acc, wm, ya, lane);
            else gemm2<L::OFF_W0A>(acc, wm, q < 2 ? ya[0] : 0.f, lane);
            if constexpr (CB == 32) gemm32<L::OFF_W0B>(acc, wm, yb, lane);
            else if constexpr (CB == 2) gemm2<L::OFF_W0B>(acc, wm, q < 2 ? yb[0] : 0.f, lane);
            float hid[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) hid[s] = relu1(acc[s >> 2][s & 3]);
            load_bias(acc, wm + L::BIAS_F, 1, q);
            gemm32<L::OFF_W1>(acc, wm, hid, lane);
#pragma unroll
            for (int s = 0; s < 8; ++s) hid[s] = relu1(acc[s >> 2][s & 3]);
            load_bias(acc, wm + L::BIAS_F, 2, q);
            gemm32<L::OFF_W2>(acc, wm, hid, lane);
            // epilogue: mask, store z, transpose through LDS, per-half {mean, M2} with lane = channel
            float v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = acc[s >> 2][s & 3];
            if (!full) {
#pragma unroll
                for (int s = 0; s < 8; ++s) v[s] = valid ? v[s] : 0.f;
            }
            store8(v, vz[m], lane_voff(vz[m], q, p, inb), g * vz[m].gs4);
            stage8(tl, lane_base, v);
            // lane (ch, h2) owns pixels 8 h2 .. 8 h2 + 7 of channel ch
            const int ch = lane & 31, h2 = lane >> 5;
            const float4 *rp = reinterpret_cast<const float4 *>(tl + ch * TLD + 8 * h2);
            const float4 q0 = rp[0], q1 = rp[1];
            float sum = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w));
            sum += __shfl_xor(sum, 32);
            const float mean = sum * inv;
            float m2;
            if (full) {
                const float d0 = q0.x - mean, d1 = q0.y - mean, d2 = q0.z - mean, d3 = q0.w - mean;
                const float d4 = q1.x - mean, d5 = q1.y - mean, d6 = q1.z - mean, d7 = q1.w - mean;
                m2 = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
            } else {
                const unsigned mh = vmask >> (8 * h2);
                const float d0 = (mh & 1u) ? q0.x - mean : 0.f, d1 = (mh & 2u) ? q0.y - mean : 0.f;
                const float d2 = (mh & 4u) ? q0.z - mean : 0.f, d3 = (mh & 8u) ? q0.w - mean : 0.f;
                const float d4 = (mh & 16u) ? q1.x - mean : 0.f, d5 = (mh & 32u) ? q1.y - mean : 0.f;
                const float d6 = (mh & 64u) ? q1.z - mean : 0.f, d7 = (mh & 128u) ? q1.w - mean : 0.f;
                m2 = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
            }
            m2 += __shfl_xor(m2, 32);
            if (lane < 32) reinterpret_cast<float2 *>(A.part[m])[((long long)g * hpg + hh) * FGNN_H + row_chan(lane)] = make_float2(mean, m2);
        }
        if (lane == 0) A.cnt[(long long)g * hpg + hh] = cnt;
        h = hn;
    }
    // halves this wave stepped over: empty records (z of a half past the plane does not exist; of a padding-only tile it is not
    // written -- every consumer of a ragged slab steps over the same tiles or reads the valid corner only)
    for (int t = H0 + wv; t < H1; t += NW) {
        const int g = t / hpg, hh = t - g * hpg;
        bool live = hh * 16 < P2;
        if constexpr (SKIP) live = live && tile_live(hh >> 1, A.N, A.nvalid[g]);
        if (live) continue;
#pragma unroll
        for (int m = 0; m < NMLP; ++m) {
            if (lane < 32) reinterpret_cast<float2 *>(A.part[m])[((long long)g * hpg + hh) * FGNN_H + lane] = make_float2(0.f, 0.f);
        }
        if (lane == 0) A.cnt[(long long)g * hpg + hh] = 0.f;
    }
}

template <int CA, int CB, int NMLP, bool PKD, bool SKIP>
int launch_fwd16(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st) {
    using L = FwdLayout16<CA, CB, NMLP>;
    constexpr int LDS = L::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_fwd_t16_kernel<CA, CB, NMLP, PKD, SKIP>, LDS);
    int grid = (2 * total + L::NW - 1) / L::NW;
    const int cap = a->cu_share == 2 ? 128 : 256;
    if (grid > cap) grid = cap;
    if (SKIP) grid = FGNN_RANGE_WG;
    hipLaunchKernelGGL((mlp_fwd_t16_kernel<CA, CB, NMLP, PKD, SKIP>), dim3(grid), dim3(64 * L::NW), LDS, st, *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CA, int CB, int NMLP, bool PKD>
int launch_fwd16s(const fgnn_mlp_fwd_args *a, int tpg, int total, hipStream_t st) {
    return a->ranges ? launch_fwd16<CA, CB, NMLP, PKD, true>(a, tpg, total, st) : launch_fwd16<CA, CB, NMLP, PKD, false>(a, tpg, total, st);
}

}  // namespace

// statistics records per graph written by fgnn_mlp_fwd_t16 (16-pixel halves; the last one of a graph may be empty)
extern "C" int fgnn_mlp_fwd_t16_records(int N) { return 2 * fgnn_tiles_per_graph(N); }

extern "C" int fgnn_mlp_fwd_t16_supported(const fgnn_mlp_fwd_args *a) {
    if (!a || a->depth != 3 || !a->packed || a->N > 256) return 0;
    const int ca = a->a.C, cb = a->b.C;
    if (a->nmlp == 2) return (cb == 0 && (ca == 32 || ca == 2)) ? 1 : 0;
    if (a->nmlp == 1) return (ca == 32 && (cb == 32 || cb == 2) && !(cb == 2 && a->b.nrm)) ? 1 : 0;
    return 0;
}

// Same contract as fgnn_mlp_fwd, except: `packed` is an image of kind 4, and part[m] / cnt hold fgnn_mlp_fwd_t16_records(N) records per graph.
extern "C" int fgnn_mlp_fwd_t16(const fgnn_mlp_fwd_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_fwd_t16: null args");
    FGNN_CHECK(fgnn_mlp_fwd_t16_supported(a), "fgnn_mlp_fwd_t16: built for depth 3, an operand image of kind 4, N <= 256 and inputs of 32 or 2 "
               "channels (two MLPs) or 32 + 32 / 32 + 2 channels (one MLP); use fgnn_mlp_fwd");
    FGNN_CHECK(a->G > 0 && a->N > 0, "fgnn_mlp_fwd_t16: bad G=%d N=%d", a->G, a->N);
    const bool pk_a = a->xbits && a->a.C == 2, pk_b = a->xbits && a->b.C == 2;
    FGNN_CHECK(a->a.ptr || pk_a, "fgnn_mlp_fwd_t16: slab a missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr || pk_b, "fgnn_mlp_fwd_t16: slab b has channels but no pointer");
    FGNN_CHECK(!a->xbits || a->xdeg, "fgnn_mlp_fwd_t16: xbits without xdeg (fgnn_adjacency_degree)");
    FGNN_CHECK(!(a->a.C == 2 && a->a.nrm), "fgnn_mlp_fwd_t16: the 2-channel slab is the raw model input");
    FGNN_CHECK((long long)a->N * a->N <= a->ldz && (pk_a || (long long)a->N * a->N <= a->a.ldp), "fgnn_mlp_fwd_t16: channel stride < N*N");
    for (int m = 0; m < a->nmlp; ++m) FGNN_CHECK(a->z[m] && a->part[m], "fgnn_mlp_fwd_t16: missing output %d", m);
    FGNN_CHECK(a->cnt, "fgnn_mlp_fwd_t16: missing cnt");
    FGNN_CHECK(!a->ranges || a->nvalid, "fgnn_mlp_fwd_t16: ranges (fgnn_ragged_tile_ranges) only make sense with nvalid");
    {
        const long long lim = 0x7fffffffll / 4, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * FGNN_H * a->ldz < lim,
                   "fgnn_mlp_fwd_t16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph(a->N);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 29), "fgnn_mlp_fwd_t16: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    const int ca = a->a.C, cb = a->b.C;
    if (a->nmlp == 2) {
        if (ca == 32) return launch_fwd16s<32, 0, 2, false>(a, tpg, (int)total, st);
        return a->xbits ? launch_fwd16s<2, 0, 2, true>(a, tpg, (int)total, st) : launch_fwd16s<2, 0, 2, false>(a, tpg, (int)total, st);
    }
    if (cb == 32) return launch_fwd16s<32, 32, 1, false>(a, tpg, (int)total, st);
    return a->xbits ? launch_fwd16s<32, 2, 1, true>(a, tpg, (int)total, st) : launch_fwd16s<32, 2, 1, false>(a, tpg, (int)total, st);
}
