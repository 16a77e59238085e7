// bf16 variant of the fused MlpBlock_Real backward (autograd of models/layers.py:126-131 with the GraphNorm backward of
// :68-80 folded into the load of dz) for gfx950.  Storage bf16, matrix work on v_mfma_f32_32x32x16_bf16, fp32 accumulation of
// every parameter gradient.  See mlp_fwd16.hip for the tile / fragment conventions (64-element tiles = 32 pixel pairs, two
// 32-column problems E / O per tile).
//
// Per tile and per pixel group:
//   1. recompute the hidden activations h_0 .. h_{d-2} with the forward's in-register chain (bf16 operands);
//   2. dz = R(ca*dy + cb*(z - mean) + cc)                                  (coef from fgnn_gn_bwd_coef*)
//   3. for l = d-1 .. 0:  dW_l += dpre_l (x) in_l,  db_l += dpre_l,  d in_l = R(W_l)^T dpre_l,
//                         dpre_{l-1} = R(d in_l * [h_{l-1} > 0])
//   4. dx = R(d in_0 (+ old dx)), optionally the per-tile sums {sum dx, sum dx (z_in - mean_in)} of the rounded values.
// The weight-gradient products contract over pixels and need lane = channel operands.  They are NOT staged through LDS:
// a fragment is transposed by multiplying it with an identity matrix on the (otherwise idle) matrix pipe
// (fgnn_bf16.h: transpose16, exact), which also yields the bias gradients as register sums.
// dW/db accumulate in registers over the wave's statically assigned tiles; the eight waves of a workgroup are summed through
// LDS in a fixed order and one partial per workgroup is written for fgnn_grad_finalize: bit-reproducible run to run.
#include <type_traits>
#include "fgnn_bf16.h"

namespace {

#ifndef FGNN_NWB
#define FGNN_NWB 8
#endif
constexpr int NWB = FGNN_NWB;    // waves per workgroup (2 per SIMD)
constexpr int BWD16_WG = 256;   // persistent workgroups (partials layout shared with the fp32 path)

template <int CA, int CB, int DEPTH>
struct Bwd16Layout {
    static constexpr Pk16 PK = pk16_layout(1, CA, CB, DEPTH);
    static constexpr int WEIGHT_F = PK.floats;
    static constexpr int REC_F = 64 + 64 + 128;                     // per wave: {a, b'} slab a, slab b, {mean, ca, cb, cc}
    static constexpr int PCOUNT = 32 * (CA + CB) + 32 + (DEPTH - 1) * (32 * 32 + 32);
    static constexpr int MAIN_F = WEIGHT_F + NWB * REC_F;
    // Weight-gradient accumulator tiles kept in LDS between the tiles of the loop ("parked") instead of in registers: the
    // variants that would otherwise spill them to scratch (the 64-input-channel kernel needs four 32x32 fp32 accumulators on
    // top of everything else).  A scratch reload retires in order with the prefetch loads in flight and stalls behind them;
    // LDS does not, and ~130 KB of it are idle here.  Slots in order of use: dW_2, dW_1, dW_0 (slab a), dW_0 (slab b).
    static constexpr int NPARK = (CA >= 32 && CB >= 32) ? 4 : (CA >= 32 && CB > 0) ? 2 : (CA >= 32 ? 1 : 0);
    static constexpr int PARK_OFF = (MAIN_F + 3) & ~3;
    static constexpr int PARK_F = NWB * NPARK * 1024;
    static constexpr int RED_F = NWB * PCOUNT;
    static constexpr int LDS_F = PARK_OFF + PARK_F > RED_F ? PARK_OFF + PARK_F : RED_F;
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

// (the base pointers pass through an empty asm: the per-lane 64-bit addresses of these once-per-graph loads are then formed here, in
// the rarely taken branch, instead of being hoisted out of the tile loop -- where they were the kernel's only spilled registers)
template <typename T>
DEVI const T *opaque_ptr(const T *p) {
    asm volatile("" : "+s"(p));
    return p;
}
DEVI void fetch_rec2(float *rec, const fgnn_slab16 &s, int g, int lane) {
    if (lane < 32) {
        float2 o = make_float2(1.f, 0.f);
        if (s.nrm && lane < s.C) {
            const float4 n = reinterpret_cast<const float4 *>(opaque_ptr(s.nrm))[(long long)g * s.C + lane];
            const float be = s.beta ? opaque_ptr(s.beta)[lane] : 0.f;
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
template <int CA, int CB, int DEPTH, bool SKIP = false>
__global__ __launch_bounds__(64 * NWB, NWB / 4) void mlp_bwd16_kernel(const fgnn_mlp_bwd16_args A, const int tpg,
                                                                 const int total_tiles) {
    static_assert(DEPTH == 3, "built for depth_of_mlp = 3");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = Bwd16Layout<CA, CB, DEPTH>;
    constexpr Pk16 PK = L::PK;
    constexpr int CIN = CA + CB, SA = pk16_steps(CA), SB = pk16_steps(CB);
    constexpr int XA = CA >= 32 ? 16 : 2, XB = CB >= 32 ? 16 : 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    young_prio(0, wv, blockDim.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int PP = A.N * A.ldr;
    const View16 va = make_view16(A.a.ptr, A.a.gstride, A.a.ldp, A.G);
    const View16 vb = make_view16(A.b.ptr, A.b.gstride, A.b.ldp, A.G);
    const View16 vdy = make_view16(A.dy, A.dgstride, A.ldd, A.G);
    const View16 vz = make_view16(A.z, A.zgstride, A.ldz, A.G);
    const View16 vdxa = make_view16(A.dxa, A.dxa_gstride, A.dxa_ld, A.G);
    const View16 vdxb = make_view16(A.dxb, A.dxb_gstride, A.dxb_ld, A.G);

    float *wl = smem;
    const float *tail = wl + PK.bias_f;
    float *rec = smem + L::WEIGHT_F + wv * L::REC_F;
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
    const bool emit = (CA == 32) && (CB == 0 ? normA : !normA) && A.dxa != nullptr && A.s12part != nullptr;

    {
        const float4 *src = reinterpret_cast<const float4 *>(A.packed);
        float4 *dst = reinterpret_cast<float4 *>(wl);
        for (int e = threadIdx.x; e < L::WEIGHT_F / 4; e += 64 * NWB) dst[e] = src[e];
    }
    unsigned xa[XA], xb[CB > 0 ? XB : 1];
    int cached_g = -1, cur_nv = A.N;
    float la_a = 1.f, la_b = 0.f, lb_a = 1.f, lb_b = 0.f, la_mean = 0.f;      // lane-channel constants (transposed layout)
    auto graph_change = [&](int g) {
        fetch_rec2(recA, A.a, g, lane);
        if constexpr (CB > 0) fetch_rec2(recB, A.b, g, lane);
        if (lane < 32) reinterpret_cast<float4 *>(recK)[lane] = reinterpret_cast<const float4 *>(opaque_ptr(A.coef))[(long long)g * FGNN_H + lane];
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
        if (normA) la_mean = opaque_ptr(A.a.nrm)[((long long)g * A.a.C + j) * 4];
    };
    int first = T0 + wv;
    if constexpr (SKIP) first = __builtin_amdgcn_readfirstlane(next_live_tile_p(first, T1, NWB, tpg, 64, A.ldr, A.nvalid));
    {
        const int t = first;
        const Tile16 c = decode16(t, t < T1, tpg, A.ldr, PP, j);
        load_slab16<CA>(xa, va, c, h);
        if constexpr (CB > 0) load_slab16<CB>(xb, vb, c, h);
        if (t < T1) graph_change(c.g);
    }
    __syncthreads();

    // row offsets of the 16 channel rows a lane touches (all 32-channel tensors of a launch share one channel stride)
    const int ld2 = va.ld2;
    auto roff = [&](int r) { return ((r & 3) + 8 * (r >> 2)) * ld2; };

    int tnext = 0;
    for (int tile = first; tile < T1; tile = tnext) {
        tnext = tile + NWB;
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
        const bool rmw_a = (CA >= 32) && A.dxa != nullptr && A.accumulate_a;
        const bool rmw_b = (CB >= 32) && A.dxb != nullptr && A.accumulate_b;
        if constexpr (CA >= 32) {
            if (rmw_a) {
                const int vo = lo4 + c.g * vdxa.gs2;
#pragma unroll
                for (int r = 0; r < 16; ++r) olda[r] = buf_load_u32(vdxa, vo, roff(r));
            }
        }
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

            // ---- dx of slab a ----
            if constexpr (CA >= 32) {
                if (A.dxa) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    f32x16 acc = mfma16(lds_step(wl, PK.off_wt0a, lane), step_of(d, 0), zero);
                    acc = mfma16(lds_step(wl, PK.off_wt0a + 1, lane), step_of(d, 1), acc);
                    if (rmw_a) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += half(olda[r]);
                    }
                    F16 v;
                    pack_acc(v, acc);
                    if constexpr (CB == 0) {
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
                    }
                    if constexpr (CB > 0) {
                        if (emit) {          // un-normalised slab: yTa is the transposed raw input itself
                            const f32x16 tv = transpose16(v, ident);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                es1 += tv[2 * q] + tv[2 * q + 1];
                                es2 = fmaf(tv[2 * q], bf_lo(yTa.d[q]), es2);
                                es2 = fmaf(tv[2 * q + 1], bf_hi(yTa.d[q]), es2);
                            }
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
            if constexpr (CB == 0) {
                if (h == 0) reinterpret_cast<float2 *>(A.s12part)[((long long)c.g * FGNN_H + j) * tpg + c.tt] = make_float2(es1, es2);
            } else {
                // the trace term only, (G, 32, tpg): its one reader (the matmul workgroup of (g, c)) walks the tiles of one channel
                if (h == 0) A.s12part[((long long)c.g * FGNN_H + j) * tpg + c.tt] = es2;
            }
        }
        // the wave's next tile
        {
            const int tn = tnext;
            const Tile16 cn = decode16(tn, tn < T1, tpg, A.ldr, PP, j);
            load_slab16<CA>(xa, va, cn, h);
            if constexpr (CB > 0) load_slab16<CB>(xb, vb, cn, h);
        }
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
    float4 *out = reinterpret_cast<float4 *>(A.wpart + (long long)blockIdx.x * PCOUNT);
    const float4 *part4 = reinterpret_cast<const float4 *>(smem);
    for (int e = threadIdx.x; e < PCOUNT / 4; e += 64 * NWB) {
        float4 a = part4[e];
#pragma unroll
        for (int w = 1; w < NWB; ++w) {                                  // fixed order
            const float4 b = part4[w * (PCOUNT / 4) + e];
            a.x += b.x;
            a.y += b.y;
            a.z += b.z;
            a.w += b.w;
        }
        out[e] = a;
    }
}

template <int CA, int CB, int DEPTH, bool SKIP>
int launch_bwd16_impl(const fgnn_mlp_bwd16_args *a, int tpg, int total, hipStream_t st) {
    constexpr int LDS = Bwd16Layout<CA, CB, DEPTH>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp_bwd16_kernel<CA, CB, DEPTH, SKIP>, LDS);
    hipLaunchKernelGGL((mlp_bwd16_kernel<CA, CB, DEPTH, SKIP>), dim3(BWD16_WG), dim3(64 * NWB), LDS, st, *a, tpg, total);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int CA, int CB, int DEPTH>
int launch_bwd16(const fgnn_mlp_bwd16_args *a, int tpg, int total, hipStream_t st) {
    static_assert(BWD16_WG == FGNN_RANGE_WG, "fgnn_ragged_tile_ranges16 splits for the backward grid");
    if (a->ranges) return launch_bwd16_impl<CA, CB, DEPTH, true>(a, tpg, total, st);
    return launch_bwd16_impl<CA, CB, DEPTH, false>(a, tpg, total, st);
}

}  // namespace

extern "C" int fgnn_mlp_bwd16(const fgnn_mlp_bwd16_args *a, void *stream) {
    FGNN_CHECK(a != nullptr, "fgnn_mlp_bwd16: null args");
    FGNN_CHECK(a->G > 0 && a->N > 0 && a->ldr >= a->N && a->ldr % 8 == 0, "fgnn_mlp_bwd16: bad G=%d N=%d ldr=%d", a->G, a->N, a->ldr);
    FGNN_CHECK(a->depth == 3, "fgnn_mlp_bwd16: built for depth_of_mlp = 3 (got %d)", a->depth);
    FGNN_CHECK(a->a.ptr && a->a.C > 0 && a->packed, "fgnn_mlp_bwd16: slab a / operand image missing");
    FGNN_CHECK(a->b.C == 0 || a->b.ptr, "fgnn_mlp_bwd16: slab b has channels but no pointer");
    FGNN_CHECK(a->dy && a->z && a->wpart && a->coef, "fgnn_mlp_bwd16: missing dy/z/wpart/coef");
    FGNN_CHECK(!a->s12part || (a->a.C == 32 && a->dxa && ((a->b.C == 0 && a->a.nrm) || (a->b.C > 0 && !a->a.nrm))),
               "fgnn_mlp_bwd16: s12part needs dxa and either a single normalised 32-channel slab or a raw first slab of a two-slab MLP");
    {
        const long long lim = 0x7fffffffll / 2, G = a->G;
        FGNN_CHECK(G * a->a.gstride < lim && G * a->b.gstride < lim && G * a->dgstride < lim && G * a->zgstride < lim &&
                   G * a->dxa_gstride < lim && G * a->dxb_gstride < lim,
                   "fgnn_mlp_bwd16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    }
    const int tpg = fgnn_tiles_per_graph16(a->N, a->ldr);
    const long long total = (long long)a->G * tpg;
    FGNN_CHECK(total < (1ll << 30), "fgnn_mlp_bwd16: too many tiles");
    hipStream_t st = (hipStream_t)stream;
    const int ca = a->a.C, cb = a->b.C;
    if (ca == 2 && cb == 0) return launch_bwd16<2, 0, 3>(a, tpg, (int)total, st);
    if (ca == 32 && cb == 0) return launch_bwd16<32, 0, 3>(a, tpg, (int)total, st);
    if (ca == 32 && cb == 2) return launch_bwd16<32, 2, 3>(a, tpg, (int)total, st);
    if (ca == 32 && cb == 32) return launch_bwd16<32, 32, 3>(a, tpg, (int)total, st);
    fgnn_set_error("fgnn_mlp_bwd16: unsupported input channels (%d + %d); built for 2, 32, 32+2, 32+32", ca, cb);
    return 1;
}
