// mlp1 + mlp2 of one FGNN block (models/blocks_emb.py:16-27: two MlpBlock_Real on the SAME input), bf16 backward in ONE
// launch -- the bf16 twin of mlp_bwd_pair.hip.  Per MLP and per tile the algorithm, rounding points and fragment conventions
// are mlp_bwd16.hip's; what changes is who does it and what travels through HBM:
//   * the two waves that share a SIMD form a PAIR on the same tile: wave p (p = 0..3) runs mlp1, wave p + 4 runs mlp2, each with
//     only its own weight-gradient accumulators (registers + the same parked LDS tiles as the single-MLP kernel);
//   * the gradient of the shared input is summed in the pair: the mlp1 wave leaves its fp32 dx fragments (even / odd pixel
//     group) in a double-buffered LDS slot, the mlp2 wave forms R(R(d_in3 + dx1) + dx2) -- the two roundings of the two
//     read-modify-write launches it replaces, so the stored d_in is bit-identical to theirs -- and stores once.  Two LDS words
//     per pair (release / acquire at workgroup scope) order the hand-over; with two buffers neither wave waits in steady state.
//   Per block: x is read once, d_in is read and written once (7 slab passes instead of 10: 287 MB instead of 422 MB at the
//   cfg4 size), one prologue / tail instead of two.
// Depth 3, one input slab of 32 channels (blocks > 1) or 2 channels (block 1: no input gradient, the pair only shares the
// launch), constant-size batches.
#include <type_traits>
#include "fgnn_bf16.h"

namespace {

constexpr int NWB = 8;           // waves per workgroup: 4 pairs
constexpr int NP = 4;
constexpr int BWD16_WG = 256;   // persistent workgroups (partials layout shared with the fp32 path)

template <int CA, int CB, int DEPTH>
struct Bwd16Layout {
    static constexpr Pk16 PK = pk16_layout(1, CA, CB, DEPTH);
    static constexpr int WEIGHT_F = PK.floats;
    static constexpr int REC_F = 64 + 64 + 128;                     // per wave: {a, b'} slab a, slab b, {mean, ca, cb, cc}
    static constexpr int PCOUNT = 32 * (CA + CB) + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int MAIN_F = 2 * WEIGHT_F + NWB * REC_F;       // two operand images
    // Weight-gradient accumulator tiles kept in LDS between the tiles of the loop ("parked") instead of in registers: the
    // variants that would otherwise spill them to scratch (the 64-input-channel kernel needs four 32x32 fp32 accumulators on
    // top of everything else).  A scratch reload retires in order with the prefetch loads in flight and stalls behind them;
    // LDS does not, and ~130 KB of it are idle here.  Slots in order of use: dW_2, dW_1, dW_0 (slab a), dW_0 (slab b).
    static constexpr int NPARK = (CA >= 32 && CB >= 32) ? 4 : (CA >= 32 && CB > 0) ? 2 : (CA >= 32 ? 1 : 0);
    static constexpr int PARK_OFF = (MAIN_F + 3) & ~3;
    static constexpr int PARK_F = NWB * NPARK * 1024;
    // hand-over of the mlp1 wave's fp32 dx fragments: per pair 2 buffers x 2 pixel groups x [4][64 lanes][4 floats]
    static constexpr int XCH_OFF = PARK_OFF + PARK_F;
    static constexpr int XCH_F = (CA >= 32) ? NP * 2 * 2 * 1024 : 0;
    static constexpr int FLAG_OFF = XCH_OFF + XCH_F;
    static constexpr int RED_F = NWB * PCOUNT;
    static constexpr int LDS_F = FLAG_OFF + 4 * NP > RED_F ? FLAG_OFF + 4 * NP : RED_F;
};

// a parked accumulator tile: [4][64 lanes][4 floats] -> conflict-free 16-byte accesses
DEVI f32x16 park_get(const float *slot, int lane) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = reinterpret_cast<const float4 *>(slot)[q * 64 + lane];
        v[4 * q] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
    }
    return v;
}
DEVI void park_put(float *slot, int lane, const f32x16 &v) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        reinterpret_cast<float4 *>(slot)[q * 64 + lane] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

DEVI void fetch_rec2(float *rec, const fgnn_slab16 &s, int g, int lane) {
    if (lane < 32) {
        float2 o = make_float2(1.f, 0.f);
        if (s.nrm && lane < s.C) {
            const float4 n = reinterpret_cast<const float4 *>(s.nrm)[(long long)g * s.C + lane];
            const float be = s.beta ? s.beta[lane] : 0.f;
            o.x = n.y;
            o.y = be - n.x * n.y;
        }
        reinterpret_cast<float2 *>(rec)[lane] = o;
    }
}

// normal fragments (even / odd pixel) of a 32-channel slab, normalised; `raw*` = the un-normalised fragments
DEVI void operands32b(F16 &e, F16 &o, F16 &rawE, F16 &rawO, const unsigned (&x)[16], const float *rec, bool norm, int h) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        rawE.d[q] = pack_lo(x[2 * q], x[2 * q + 1]);
        rawO.d[q] = pack_hi(x[2 * q], x[2 * q + 1]);
    }
    if (norm) {
        const float2 *r2 = reinterpret_cast<const float2 *>(rec);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 n0 = r2[ch_of(2 * q, h)], n1 = r2[ch_of(2 * q + 1, h)];
            e.d[q] = cvt_pk(fmaf(bf_lo(x[2 * q]), n0.x, n0.y), fmaf(bf_lo(x[2 * q + 1]), n1.x, n1.y));
            o.d[q] = cvt_pk(fmaf(bf_hi(x[2 * q]), n0.x, n0.y), fmaf(bf_hi(x[2 * q + 1]), n1.x, n1.y));
        }
    } else {
        e = rawE;
        o = rawO;
    }
}
DEVI void operands2b(F16 &e, F16 &o, const unsigned (&x)[2]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) e.d[q] = o.d[q] = 0u;
    e.d[0] = pack_lo(x[0], x[1]);
    o.d[0] = pack_hi(x[0], x[1]);
}

// transposed, normalised operand of one pixel group: lane = channel, y^T = R(x^T * a_lane + b_lane)
DEVI F16 transposed_input(const F16 &raw, const F16 &ident, bool norm, float la, float lb) {
    const f32x16 t = transpose16(raw, ident);
    F16 f;
    if (norm) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f.d[q] = cvt_pk(fmaf(t[2 * q], la, lb), fmaf(t[2 * q + 1], la, lb));
    } else {
        pack_acc(f, t);
    }
    return f;
}

DEVI float sum16(const f32x16 &t) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += t[r];
    return s;
}

// SKIP (ragged batches with A.ranges): work-balanced tile range from fgnn_ragged_tile_ranges16; the waves step over tiles
// without a valid element (no contribution to the parameter gradients, dx not written there); such a tile only gets an empty
// S1/S2 (or trace-term) record.
struct Pair16Args {
    fgnn_mlp_bwd16_args m[2];
};

template <int CA>
__global__ __launch_bounds__(64 * NWB, NWB / 4) void mlp_bwd16_pair_kernel(const Pair16Args P, const int tpg, const int total_tiles) {
    constexpr int CB = 0, DEPTH = 3;
    constexpr bool SKIP = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Bwd16Layout<CA, CB, DEPTH>;
    constexpr Pk16 PK = L::PK;
    constexpr int CIN = CA + CB, SA = pk16_steps(CA), SB = pk16_steps(CB);
    constexpr int XA = CA >= 32 ? 16 : 2, XB = CB >= 32 ? 16 : 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wv >> 2, pair = wv & 3;        // role 0: mlp1 (hands its dx over), role 1: mlp2 (sums, stores, emits)
    const fgnn_mlp_bwd16_args &A = P.m[role];
    const int j = lane & 31, h = lane >> 5;
    const int PP = A.N * A.ldr;
    const View16 va = make_view16(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View16 vdy = make_view16(A.dy, A.dgstride, A.ldd, A.G);
    const View16 vz = make_view16(A.z, A.zgstride, A.ldz, A.G);
    const View16 vdxa = make_view16(P.m[1].dxa, P.m[1].dxa_gstride, P.m[1].dxa_ld, P.m[1].G);
    const View16 vdxb = vdxa;      // (unused: single-slab MLPs)

    float *wl = smem + role * L::WEIGHT_F;
    const float *tail = wl + PK.bias_f;
    float *rec = smem + 2 * L::WEIGHT_F + wv * L::REC_F;
    float *xch = smem + L::XCH_OFF + pair * (2 * 2 * 1024);
    int *flags = reinterpret_cast<int *>(smem + L::FLAG_OFF) + 4 * pair;     // [0] = last tile handed over, [1] = last tile consumed
    float *recA = rec, *recB = rec + 64, *recK = rec + 128;
    const F16 ident = make_identity(lane);

    f32x16 dW0a, dW0b, dWh[DEPTH - 1];
    float db[DEPTH];
    zero16f(dW0a);
    zero16f(dW0b);
#pragma unroll
    for (int l = 0; l + 1 < DEPTH; ++l) zero16f(dWh[l]);
    // slot s of this wave's parked accumulators (s < NPARK), zero-initialised; `accum(slot, reg, f)` applies f to the tile
    constexpr int NPARK = L::NPARK;
    float *park = smem + L::PARK_OFF + wv * (NPARK * 1024);
#pragma unroll
    for (int s = 0; s < NPARK; ++s) park_put(park + s * 1024, lane, dW0a);
    auto accum = [&](auto slot, f32x16 &reg, auto &&f) {
        constexpr int S = decltype(slot)::value;
        if constexpr (S < NPARK) {
            f32x16 a = park_get(park + S * 1024, lane);
            f(a);
            park_put(park + S * 1024, lane, a);
        } else {
            f(reg);
        }
    };
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] = 0.f;

    const int nwg = gridDim.x;
    const int q_ = total_tiles / nwg, rem = total_tiles % nwg;
    int T0 = blockIdx.x * q_ + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
    int T1 = T0 + q_ + ((int)blockIdx.x < rem ? 1 : 0);
    if constexpr (SKIP) {
        T0 = A.ranges[blockIdx.x];
        T1 = A.ranges[blockIdx.x + 1];
    }
    const bool normA = (CA >= 32) && A.a.nrm != nullptr, normB = (CB >= 32) && A.b.nrm != nullptr;
    // per-tile sums of the slab-a gradient: for a normalised single slab {sum dx, sum dx (z_a - mean_a)} (the GraphNorm backward
    // sums of its producer); for the raw first slab of a two-slab MLP (mlp3: slab a = mult) {sum dx, sum dx * x_a} =
    // the trace term T = <dM, M> from which fgnn_chan_matmul_bwd16 derives the S2 sums of both its operands
    const bool has_dx = (CA == 32) && P.m[1].dxa != nullptr;
    const bool emit = (CA == 32) && role == 1 && normA && has_dx && P.m[1].s12part != nullptr;

    {
        float4 *dst = reinterpret_cast<float4 *>(smem);
        for (int e = threadIdx.x; e < 2 * (L::WEIGHT_F / 4); e += 64 * NWB) {
            const int m = e >= L::WEIGHT_F / 4 ? 1 : 0;
            dst[e] = reinterpret_cast<const float4 *>(P.m[m].packed)[e - m * (L::WEIGHT_F / 4)];
        }
        if (threadIdx.x < 4 * NP) reinterpret_cast<int *>(smem + L::FLAG_OFF)[threadIdx.x] = -1;
    }
    unsigned xa[XA], xb[CB > 0 ? XB : 1];
    int cached_g = -1, cur_nv = A.N;
    float la_a = 1.f, la_b = 0.f, lb_a = 1.f, lb_b = 0.f, la_mean = 0.f;      // lane-channel constants (transposed layout)
    auto graph_change = [&](int g) {
        fetch_rec2(recA, A.a, g, lane);
        if constexpr (CB > 0) fetch_rec2(recB, A.b, g, lane);
        if (lane < 32) reinterpret_cast<float4 *>(recK)[lane] = reinterpret_cast<const float4 *>(A.coef)[(long long)g * FGNN_H + lane];
        cached_g = g;
        cur_nv = __builtin_amdgcn_readfirstlane(nvalid_of(A.nvalid, g, A.N));
        const float2 ra = reinterpret_cast<const float2 *>(recA)[j];
        la_a = ra.x;
        la_b = ra.y;
        if constexpr (CB > 0) {
            const float2 rb = reinterpret_cast<const float2 *>(recB)[j];
            lb_a = rb.x;
            lb_b = rb.y;
        }
        if (normA) la_mean = A.a.nrm[((long long)g * A.a.C + j) * 4];
    };
    int first = T0 + pair;
    if constexpr (SKIP) first = __builtin_amdgcn_readfirstlane(next_live_tile_p(first, T1, NWB, tpg, 64, A.ldr, A.nvalid));
    {
        const int t = first;
        const Tile16 c = decode16(t, t < T1, tpg, A.ldr, PP, j);
        load_slab16<CA>(xa, va, c, h);
        if (t < T1) graph_change(c.g);
    }
    __syncthreads();

    // row offsets of the 16 channel rows a lane touches (all 32-channel tensors of a launch share one channel stride)
    const int ld2 = va.ld2;
    auto roff = [&](int r) { return ((r & 3) + 8 * (r >> 2)) * ld2; };

#ifndef FGNN_XEARLY
#define FGNN_XEARLY 0     // measurement switch: x of the next tile requested in the middle of the tile (10 spilled registers: 75 -> 87 us)
#endif
#ifndef FGNN_PRIO16
#define FGNN_PRIO16 1
#endif
    // static priority for the mlp2 waves (the younger half of the workgroup AND the longer half of the pair): see mlp_bwd_pair_t16.hip
    if (FGNN_PRIO16 > 0 && role == 1) __builtin_amdgcn_s_setprio(FGNN_PRIO16);
    int tnext = 0;
    for (int tile = first; tile < T1; tile = tnext) {
        tnext = tile + NP;
        if constexpr (SKIP) tnext = __builtin_amdgcn_readfirstlane(next_live_tile_p(tnext, T1, NWB, tpg, 64, A.ldr, A.nvalid));
        const Tile16 c = decode16(tile, true, tpg, A.ldr, PP, j);
        if (c.g != cached_g) graph_change(c.g);
        const bool v0 = c.inb && c.i < cur_nv && c.jj < cur_nv;
        const bool v1 = c.inb && c.i < cur_nv && c.jj + 1 < cur_nv;
        const int lo4 = c.inb ? 4 * h * ld2 + 4 * c.pp : OOB_OFF;      // lane part of every 32-channel access

        // ---- all loads of the tile are requested up front ----
        unsigned dyr[16], zr[16];
        {
            const int vo_dy = lo4 + c.g * vdy.gs2, vo_z = lo4 + c.g * vz.gs2;
#pragma unroll
            for (int r = 0; r < 16; ++r) dyr[r] = buf_load_u32(vdy, vo_dy, roff(r));
#pragma unroll
            for (int r = 0; r < 16; ++r) zr[r] = buf_load_u32(vz, vo_z, roff(r));
        }
        unsigned olda[CA >= 32 ? 16 : 1], oldb[CB >= 32 ? 16 : 1];
        const bool rmw_a = has_dx && role == 1 && P.m[1].accumulate_a;
        float *xbuf = xch + (((tile - first) / NP) & 1) * (2 * 1024);          // this tile's hand-over buffer
        const bool rmw_b = (CB >= 32) && A.dxb != nullptr && A.accumulate_b;
        if constexpr (CA >= 32) {
            if (has_dx) {     // (issued on every path -- out of range when nothing is accumulated: no traffic -- so that the compiler can count
                              // what is in flight behind the x prefetch below and its s_waitcnt for x does not drain these)
                const int vo = (rmw_a ? lo4 : OOB_OFF) + c.g * vdxa.gs2;
#pragma unroll
                for (int r = 0; r < 16; ++r) olda[r] = buf_load_u32(vdxa, vo, roff(r));
            }
        }
        const Tile16 cn = decode16(tnext, tnext < T1, tpg, A.ldr, PP, j);       // the wave's next tile
        if constexpr (CB >= 32) {
            if (rmw_b) {
                const int vo = lo4 + c.g * vdxb.gs2;
#pragma unroll
                for (int r = 0; r < 16; ++r) oldb[r] = buf_load_u32(vdxb, vo, roff(r));
            }
        }
        F16 keepA, keepB;           // rounded dx of the even pixels, waiting for the odd ones
        float es1 = 0.f, es2 = 0.f; // S1 / S2 of the tile (emit)

        // One pixel group (GRP 0 = even, 1 = odd pixels of the pairs) end to end.  The two groups are separated by a
        // scheduling barrier: interleaving them doubles the live fragments and spills.
        auto group = [&](auto tag) {
            constexpr int GRP = decltype(tag)::value;
            auto half = [](unsigned d) { return GRP ? bf_hi(d) : bf_lo(d); };
            auto pack2 = [](unsigned a, unsigned b) { return GRP ? pack_hi(a, b) : pack_lo(a, b); };
            const float fv = GRP ? (v1 ? 1.f : 0.f) : (v0 ? 1.f : 0.f);
            // ---- input operands: normal (recompute) and transposed (layer-0 weight gradient) ----
            F16 ya, yb, raw_a, yTa, yTb;
            if constexpr (CA >= 32) {
#pragma unroll
                for (int q = 0; q < 8; ++q) raw_a.d[q] = pack2(xa[2 * q], xa[2 * q + 1]);
                if (normA) {
                    const float2 *r2 = reinterpret_cast<const float2 *>(recA);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float2 n0 = r2[ch_of(2 * q, h)], n1 = r2[ch_of(2 * q + 1, h)];
                        ya.d[q] = cvt_pk(fmaf(half(xa[2 * q]), n0.x, n0.y), fmaf(half(xa[2 * q + 1]), n1.x, n1.y));
                    }
                } else {
                    ya = raw_a;
                }
                yTa = transposed_input(raw_a, ident, normA, la_a, la_b);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) ya.d[q] = 0u;
                ya.d[0] = pack2(xa[0], xa[1]);
                yTa = transposed_input(ya, ident, false, 1.f, 0.f);
            }
            if constexpr (CB >= 32) {
                F16 raw_b;
#pragma unroll
                for (int q = 0; q < 8; ++q) raw_b.d[q] = pack2(xb[2 * q], xb[2 * q + 1]);
                if (normB) {
                    const float2 *r2 = reinterpret_cast<const float2 *>(recB);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float2 n0 = r2[ch_of(2 * q, h)], n1 = r2[ch_of(2 * q + 1, h)];
                        yb.d[q] = cvt_pk(fmaf(half(xb[2 * q]), n0.x, n0.y), fmaf(half(xb[2 * q + 1]), n1.x, n1.y));
                    }
                } else {
                    yb = raw_b;
                }
                yTb = transposed_input(raw_b, ident, normB, lb_a, lb_b);
            } else if constexpr (CB > 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) yb.d[q] = 0u;
                yb.d[0] = pack2(xb[0], xb[1]);
                yTb = transposed_input(yb, ident, false, 1.f, 0.f);
            }

#if FGNN_XEARLY
            // the next tile's x into the registers both groups have consumed by now: a tile's time instead of its last instants to arrive
            if constexpr (GRP == 1 && CB == 0) load_slab16<CA>(xa, va, cn, h);
#endif
            // ---- forward recompute: h_0 .. h_{d-2} ----
            F16 hs[DEPTH - 1];
            {
                f32x16 acc;
                load_bias16(acc, tail, 0, h);
#pragma unroll
                for (int t = 0; t < SA; ++t) acc = mfma16(lds_step(wl, PK.off_w0a + t, lane), step_of(ya, t), acc);
#pragma unroll
                for (int t = 0; t < SB; ++t) acc = mfma16(lds_step(wl, PK.off_w0b + t, lane), step_of(yb, t), acc);
                pack_acc_relu(hs[0], acc);
#pragma unroll
                for (int l = 1; l + 1 < DEPTH; ++l) {
                    load_bias16(acc, tail, l, h);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc = mfma16(lds_step(wl, PK.off_wh + 2 * (l - 1) + t, lane), step_of(hs[l - 1], t), acc);
                    pack_acc_relu(hs[l], acc);
                }
            }

            // ---- dz from (dy, z, coef), rounded to bf16, zero in the padding ----
            F16 d;
            {
                const float4 *kp = reinterpret_cast<const float4 *>(recK);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 k0 = kp[ch_of(2 * q, h)], k1 = kp[ch_of(2 * q + 1, h)];
                    const float e0 = fmaf(k0.y, half(dyr[2 * q]), fmaf(k0.z, half(zr[2 * q]) - k0.x, k0.w));
                    const float e1 = fmaf(k1.y, half(dyr[2 * q + 1]), fmaf(k1.z, half(zr[2 * q + 1]) - k1.x, k1.w));
                    d.d[q] = cvt_pk(e0 * fv, e1 * fv);
                }
            }

            // ---- hidden layers, l = DEPTH-1 .. 1 ----
#pragma unroll
            for (int l = DEPTH - 1; l >= 1; --l) {
                const F16 &in = hs[l - 1];
                {
                    f32x16 t = transpose16(d, ident);
                    db[l] += sum16(t);
                    F16 dT, hT;
                    pack_acc(dT, t);
                    t = transpose16(in, ident);
                    pack_acc(hT, t);
                    auto upd = [&](f32x16 &a) {
                        a = mfma16(step_of(dT, 0), step_of(hT, 0), a);
                        a = mfma16(step_of(dT, 1), step_of(hT, 1), a);
                    };
                    if (l == 2) accum(std::integral_constant<int, 0>(), dWh[l - 1], upd);
                    else accum(std::integral_constant<int, 1>(), dWh[l - 1], upd);
                }
                {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    f32x16 acc = mfma16(lds_step(wl, PK.off_wt + 2 * (DEPTH - 1 - l), lane), step_of(d, 0), zero);
                    acc = mfma16(lds_step(wl, PK.off_wt + 2 * (DEPTH - 1 - l) + 1, lane), step_of(d, 1), acc);
#pragma unroll
                    for (int q = 0; q < 8; ++q) d.d[q] = cvt_pk(acc[2 * q], acc[2 * q + 1]) & pos_mask_pk(in.d[q]);
                }
            }

            // ---- layer 0: weight gradient against the transposed inputs ----
            {
                f32x16 t = transpose16(d, ident);
                db[0] += sum16(t);
                F16 dT;
                pack_acc(dT, t);
                accum(std::integral_constant<int, 2>(), dW0a, [&](f32x16 &a) {
                    a = mfma16(step_of(dT, 0), step_of(yTa, 0), a);
                    a = mfma16(step_of(dT, 1), step_of(yTa, 1), a);
                });
                if constexpr (CB > 0) {
                    accum(std::integral_constant<int, 3>(), dW0b, [&](f32x16 &a) {
                        a = mfma16(step_of(dT, 0), step_of(yTb, 0), a);
                        a = mfma16(step_of(dT, 1), step_of(yTb, 1), a);
                    });
                }
            }

            // ---- dx of the shared input slab ----
            if constexpr (CA >= 32) {
                if (has_dx) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    f32x16 acc = mfma16(lds_step(wl, PK.off_wt0a, lane), step_of(d, 0), zero);
                    acc = mfma16(lds_step(wl, PK.off_wt0a + 1, lane), step_of(d, 1), acc);
                    if (role == 0) {
                        // hand the fp32 fragment over ([4][lane][4] like a parked tile).  The buffer was last used two tiles ago:
                        // wait until that tile has been consumed
                        if constexpr (GRP == 0) {
                            if (tile - first >= 2 * NP) {
                                while (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < tile - 2 * NP) __builtin_amdgcn_s_sleep(1);
                            }
                        }
                        park_put(xbuf + GRP * 1024, lane, acc);
                        if constexpr (GRP == 1) __hip_atomic_store(&flags[0], tile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else {
                        if constexpr (GRP == 0) {
                            while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < tile) __builtin_amdgcn_s_sleep(1);
                        }
                        // R(d_in3 + dx1) first -- what the mlp1 launch used to store -- then this MLP's share on top of it
                        f32x16 t = park_get(xbuf + GRP * 1024, lane);
                        if constexpr (GRP == 1) __hip_atomic_store(&flags[1], tile, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (rmw_a) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) t[r] += half(olda[r]);
                        }
                        F16 v1;
                        pack_acc(v1, t);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            acc[2 * q] += bf_lo(v1.d[q]);
                            acc[2 * q + 1] += bf_hi(v1.d[q]);
                        }
                        F16 v;
                        pack_acc(v, acc);
                        if (emit) {
                            // sums of the producer of slab a: S1 = sum v, S2 = sum v (z_a - mean_a), v = R(dx) (exactly 0 on
                            // invalid pixels: dz is masked and the stored padding of the old dx is 0)
                            const f32x16 tv = transpose16(v, ident), tx = transpose16(raw_a, ident);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                es1 += tv[r];
                                es2 = fmaf(tv[r], tx[r] - la_mean, es2);
                            }
                        }
                        if constexpr (GRP == 0) {
                            keepA = v;
                        } else {
                            const int vo = lo4 + c.g * vdxa.gs2;
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                buf_store_u32(pack_lo(keepA.d[q], v.d[q]), vdxa, vo, roff(2 * q));
                                buf_store_u32(pack_hi(keepA.d[q], v.d[q]), vdxa, vo, roff(2 * q + 1));
                            }
                        }
                    }
                }
            }
            // ---- dx of slab b ----
            if constexpr (CB >= 32) {
                if (A.dxb) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    f32x16 acc = mfma16(lds_step(wl, PK.off_wt0b, lane), step_of(d, 0), zero);
                    acc = mfma16(lds_step(wl, PK.off_wt0b + 1, lane), step_of(d, 1), acc);
                    if (rmw_b) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += half(oldb[r]);
                    }
                    F16 v;
                    pack_acc(v, acc);
                    if constexpr (GRP == 0) {
                        keepB = v;
                    } else {
                        const int vo = lo4 + c.g * vdxb.gs2;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            buf_store_u32(pack_lo(keepB.d[q], v.d[q]), vdxb, vo, roff(2 * q));
                            buf_store_u32(pack_hi(keepB.d[q], v.d[q]), vdxb, vo, roff(2 * q + 1));
                        }
                    }
                }
            }
        };
        group(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        group(std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (emit) {
            es1 += __shfl_xor(es1, 32);
            es2 += __shfl_xor(es2, 32);
            if (h == 0) reinterpret_cast<float2 *>(P.m[1].s12part)[((long long)c.g * FGNN_H + j) * tpg + c.tt] = make_float2(es1, es2);
        }
#if !FGNN_XEARLY
        load_slab16<CA>(xa, va, cn, h);     // the wave's next tile
#endif
    }

    if constexpr (SKIP) {       // padding-only tiles of this wave's share: empty S1/S2 / trace-term records
        if (emit) {
            for (int t = T0 + wv; t < T1; t += NWB) {
                const int g = __builtin_amdgcn_readfirstlane(t / tpg), tt = t - g * tpg;
                if (tile_live_p(tt, 64, A.ldr, A.nvalid[g])) continue;
                if (h == 0) {
                    if constexpr (CB == 0) reinterpret_cast<float2 *>(A.s12part)[((long long)g * FGNN_H + j) * tpg + tt] = make_float2(0.f, 0.f);
                    else A.s12part[((long long)g * FGNN_H + j) * tpg + tt] = 0.f;
                }
            }
        }
    }

    // ---- workgroup reduction of the parameter gradients (fixed order over the waves) ----
    // layout: [W0 (32*CIN) | b0 (32) | W1 (1024) | b1 (32) | ...]
    constexpr int PCOUNT = L::PCOUNT;
#pragma unroll
    for (int l = 0; l < DEPTH; ++l) db[l] += __shfl_xor(db[l], 32);
    // parked accumulators back into registers before the reduction buffer (which aliases them) is written
    if constexpr (NPARK > 0) dWh[1] = park_get(park, lane);
    if constexpr (NPARK > 1) dWh[0] = park_get(park + 1024, lane);
    if constexpr (NPARK > 2) dW0a = park_get(park + 2 * 1024, lane);
    if constexpr (NPARK > 3) dW0b = park_get(park + 3 * 1024, lane);
    __syncthreads();                       // everyone done with the operand image and the parked tiles
    {
        float *red = smem + wv * PCOUNT;
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
            if (h == 0) red[off + j] = db[l];
            off += 32;
        }
    }
    __syncthreads();
    static_assert(PCOUNT % 4 == 0, "partials are summed four at a time");
    const float4 *part4 = reinterpret_cast<const float4 *>(smem);
    for (int e = threadIdx.x; e < 2 * (PCOUNT / 4); e += 64 * NWB) {
        const int m = e >= PCOUNT / 4 ? 1 : 0, ee = e - m * (PCOUNT / 4);
        float4 a = part4[(4 * m) * (PCOUNT / 4) + ee];
#pragma unroll
        for (int w = 1; w < NP; ++w) {                                   // fixed order over the MLP's four waves
            const float4 b = part4[(4 * m + w) * (PCOUNT / 4) + ee];
            a.x += b.x;
            a.y += b.y;
            a.z += b.z;
            a.w += b.w;
        }
        reinterpret_cast<float4 *>(P.m[m].wpart + (long long)blockIdx.x * PCOUNT)[ee] = a;
    }
}

template <int CA>
int launch_pair16(const fgnn_mlp_bwd16_args *a1, const fgnn_mlp_bwd16_args *a2, int tpg, int total, hipStream_t st) {
    constexpr int LDS = Bwd16Layout<CA, 0, 3>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd16_pair_kernel<CA>, LDS);
    Pair16Args P;
    P.m[0] = *a1;
    P.m[1] = *a2;
    hipLaunchKernelGGL((mlp_bwd16_pair_kernel<CA>), dim3(BWD16_WG), dim3(64 * NWB), LDS, st, P, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int fgnn_mlp_bwd16_pair(const fgnn_mlp_bwd16_args *a1, const fgnn_mlp_bwd16_args *a2, void *stream) {
    FGNN_CHECK(a1 && a2, "fgnn_mlp_bwd16_pair: null args");
    FGNN_CHECK(BWD16_WG == fgnn_mlp_bwd_num_workgroups(), "fgnn_mlp_bwd16_pair: workgroup count differs from fgnn_mlp_bwd");
    FGNN_CHECK(a1->G > 0 && a1->N > 0 && a1->ldr >= a1->N && a1->ldr % 8 == 0 && a1->G == a2->G && a1->N == a2->N && a1->ldr == a2->ldr,
               "fgnn_mlp_bwd16_pair: the two MLPs must share G, N and ldr (G=%d N=%d ldr=%d)", a1->G, a1->N, a1->ldr);
    FGNN_CHECK(a1->depth == 3 && a2->depth == 3, "fgnn_mlp_bwd16_pair: built for depth_of_mlp = 3");
    FGNN_CHECK((a1->a.C == 2 || a1->a.C == 32) && a1->b.C == 0 && a2->b.C == 0,
               "fgnn_mlp_bwd16_pair: ONE input slab of 2 or 32 channels (got %d + %d); use fgnn_mlp_bwd16", a1->a.C, a1->b.C);
    FGNN_CHECK(a1->a.ptr && a1->a.ptr == a2->a.ptr && a1->a.C == a2->a.C && a1->a.gstride == a2->a.gstride && a1->a.ldp == a2->a.ldp &&
               a1->a.nrm == a2->a.nrm && a1->a.beta == a2->a.beta && a1->nvalid == a2->nvalid,
               "fgnn_mlp_bwd16_pair: the two MLPs must read the same input slab");
    FGNN_CHECK(!a1->ranges && !a2->ranges && !a1->nvalid, "fgnn_mlp_bwd16_pair: constant-size batches only; use fgnn_mlp_bwd16");
    FGNN_CHECK(a1->packed && a2->packed, "fgnn_mlp_bwd16_pair: needs both operand images (fgnn_pack16_operands, kind 1)");
    FGNN_CHECK(!a1->dxa && !a1->s12part, "fgnn_mlp_bwd16_pair: the input gradient and its tile sums belong to the SECOND argument block");
    FGNN_CHECK(!(a2->dxa && a2->a.C != 32), "fgnn_mlp_bwd16_pair: the input gradient exists for the 32-channel slab only");
    FGNN_CHECK(!a2->s12part || (a2->a.C == 32 && a2->dxa && a2->a.nrm), "fgnn_mlp_bwd16_pair: s12part needs dxa and a normalised 32-channel slab");
    for (const fgnn_mlp_bwd16_args *a : {a1, a2}) {
        FGNN_CHECK(a->dy && a->z && a->wpart && a->coef, "fgnn_mlp_bwd16_pair: missing dy/z/wpart/coef");
        const long long lim = 0x7fffffffll / 2, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim && G * a->dxa_gstride < lim,
                   "fgnn_mlp_bwd16_pair: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph16(a1->N, a1->ldr);
    const long long total = (long long)a1->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd16_pair: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    if (a1->a.C == 2) return launch_pair16<2>(a1, a2, tpg, (int)total, st);
    return launch_pair16<32>(a1, a2, tpg, (int)total, st);
}
