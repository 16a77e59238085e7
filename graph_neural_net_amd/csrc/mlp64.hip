// The conv stack of a 64-wide MlpBlock_Real (models/layers.py:113-131: depth 3, hidden = out = 64 channels, 1 .. 128 input channels
// -- the three MLPs of a 64-feature FGNN block, models/blocks_emb.py:16-36) as ONE launch forward and ONE launch
// backward on 16-pixel tiles / v_mfma_f32_16x16x4_f32 (round 6; fgnn_t16.h).  A 64-channel activation is two 32-channel fragment
// groups of the 32-wide kernels' layout, a 64 x K layer 2 x (K / 32) operand sets of 32 x 32.
//
// forward   z = W2 relu(W1 relu(W0 x + b0) + b1) + b2, zeros outside the valid corner; nothing but z is written.
// backward  recomputes the hidden activations from x (the forward's fma sequence, so the ReLU decisions are the forward's), then
//           dx = W0^T d0, dW_l += d_l (x) in_l, db_l += sum d_l with d2 = dz (masked), d1 = relu'(h2) W2^T d2, d0 = relu'(h1) W1^T d1.
//           The weight-gradient accumulators of a wave are the whole parameter set: 64 K0 / 64 + 2 x 64 registers per lane (192 at
//           K0 = 64, 256 at K0 = 128).  The kernel therefore runs ONE wave per SIMD (256 threads per CU) on the unified 512-register
//           file: the accumulators live in the AGPR half.  Per-workgroup partials, summed by fgnn_reduce_partials (fixed order).
// The operand images come packed (fgnn_mlp64_pack: one small launch per forward call; the module path hands over nn.Conv2d parameters).
#include <type_traits>
#define FGNN_STORE_AUX 0    // plain stores: z and dx are the NEXT launch's inputs (GraphNorm plane kernels); `nt` (the engine kernels' choice for
                            // their d_in) costs the 64-feature step 0.8 % (2.999 -> 2.975 ms, tools/ab_width64.sh)
#include "fgnn_t16_pipe.h"

namespace {

using namespace t16;

constexpr int OG = 2;                  // 32-channel groups of a 64-wide activation
constexpr int SUB = 1024;              // floats per 32 x 32 operand set
#ifndef M64_NWF
#define M64_NWF 8
#endif
constexpr int NWF = M64_NWF, NWB = 4;  // waves per workgroup: forward (2 per SIMD), backward (1 per SIMD)
constexpr int WG64 = 256;              // persistent workgroups = rows of wpart

DEVI int graph_nv(const rsrc_t &rnv, bool ragged, int g, int N) {
    const int v = __builtin_amdgcn_raw_buffer_load_b32(rnv, g * 4, 0, 0);
    return __builtin_amdgcn_readfirstlane(ragged ? v : N);
}

// element f of an operand set: k-step s, row block b, MFMA lane (m, kq)  (fgnn_t16.h gemm32: image element (t, lane), t = 2 s + b,
// stored [t / 4][lane][4])
struct SetElem {
    int row, col;       // row = output channel of the set (0..31), col = contracted channel (0..31)
};
DEVI SetElem set_elem(int r) {
    const int e = r & 3, lane = (r >> 2) & 63, u = r >> 8;
    const int s = 2 * u + (e >> 1), b = e & 1, m = lane & 15, kq = lane >> 4;
    SetElem o;
    o.row = chan(4 * b + (m & 3), m >> 2);
    o.col = chan(s, kq);
    return o;
}
// ---- the packed operand record of one MLP (fgnn_mlp64_pack; built once per forward call, reused by the backward) ----------------
//   [F0 | F1 | B2 | B1 | B0 | F2 | bias]     F_l: forward image of layer l, sets [og][ig]: out[32 og + row] += W[..][32 ig + col] in[..];
//   B_l: transposed image, sets [rg][kg]: din[32 rg + row] += W[32 kg + col][32 rg + row] dout[32 kg + col]; columns >= cin are zeros;
//   bias: per layer [og][q][8] = bias[32 og + chan(j, q)].  The backward copies [F0 .. B0] (or [F0 .. B1] without dx) in one piece.
struct Packed64 {
    int kg;
    DEVI constexpr int f0() const { return 0; }
    DEVI constexpr int f1() const { return OG * kg * SUB; }
    DEVI constexpr int b2() const { return f1() + OG * OG * SUB; }
    DEVI constexpr int b1() const { return b2() + OG * OG * SUB; }
    DEVI constexpr int b0() const { return b1() + OG * OG * SUB; }
    DEVI constexpr int f2() const { return b0() + kg * OG * SUB; }
    DEVI constexpr int bias() const { return f2() + OG * OG * SUB; }
    DEVI constexpr int floats() const { return bias() + 3 * 64; }
};
constexpr int packed_floats(int kg) { return (OG * kg + 3 * OG * OG + kg * OG + OG * OG) * SUB + 3 * 64; }

constexpr int PACK_MAX_JOBS = 16;
struct PackJobs {
    const float *W0[PACK_MAX_JOBS], *W1[PACK_MAX_JOBS], *W2[PACK_MAX_JOBS], *b0[PACK_MAX_JOBS], *b1[PACK_MAX_JOBS], *b2[PACK_MAX_JOBS];
    float *out[PACK_MAX_JOBS];
    int cin[PACK_MAX_JOBS];
};
// blockIdx.y = the MLP
__global__ __launch_bounds__(256) void mlp64_pack_kernel(const PackJobs J) {
    const int j = blockIdx.y;
    const float *W0 = J.W0[j], *W1 = J.W1[j], *W2 = J.W2[j], *b0 = J.b0[j], *b1 = J.b1[j], *b2 = J.b2[j];
    float *out = J.out[j];
    const int cin = J.cin[j], kg = (cin + 31) / 32;
    const Packed64 L{kg};
    const int total = L.floats();
    for (int f = blockIdx.x * 256 + threadIdx.x; f < total; f += gridDim.x * 256) {
        float v;
        if (f >= L.bias()) {
            const int r = f - L.bias(), l = r >> 6, og = (r >> 5) & 1, q = (r >> 3) & 3, jj = r & 7;
            const float *bp = l == 0 ? b0 : (l == 1 ? b1 : b2);
            v = bp ? bp[32 * og + chan(jj, q)] : 0.f;
        } else {
            // which image, forward or transposed, its weights and widths
            const float *W;
            int K, KGL, base;
            bool fwd;
            if (f < L.f1()) { W = W0; K = cin; KGL = kg; base = L.f0(); fwd = true; }
            else if (f < L.b2()) { W = W1; K = 64; KGL = OG; base = L.f1(); fwd = true; }
            else if (f < L.b1()) { W = W2; K = 64; KGL = OG; base = L.b2(); fwd = false; }
            else if (f < L.b0()) { W = W1; K = 64; KGL = OG; base = L.b1(); fwd = false; }
            else if (f < L.f2()) { W = W0; K = cin; KGL = kg; base = L.b0(); fwd = false; }
            else { W = W2; K = 64; KGL = OG; base = L.f2(); fwd = true; }
            const int r = f - base, sub = r >> 10;
            const SetElem e = set_elem(r & 1023);
            int o, c;
            if (fwd) {
                const int og = sub / KGL, ig = sub - og * KGL;
                o = 32 * og + e.row;
                c = 32 * ig + e.col;
            } else {
                const int rg = sub / OG, kgi = sub - rg * OG;
                c = 32 * rg + e.row;
                o = 32 * kgi + e.col;
            }
            v = c < K ? W[o * K + c] : 0.f;
        }
        out[f] = v;
    }
}

// n4 float4 from the packed record to LDS, all threads of the workgroup
DEVI void copy4(float *dst, const float *src, int n4, int tid, int nthr) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int e = tid; e < n4; e += nthr) d4[e] = s4[e];
}

// The kernels run one (backward) or two (forward) waves per SIMD, so nothing hides an LDS read's latency but the wave's own MFMAs:
// every GEMM helper below requests the operands of set i + 1 BEFORE issuing the 16 MFMAs of set i (LDS returns in order, the
// compiler's s_waitcnt lgkmcnt(n) then waits for set i only).
// Across helpers the chain continues: each takes the operands of its FIRST set already loaded (`first`) and calls `next()` before the
// MFMAs of its LAST set -- the caller loads the following helper's first operands there.
struct NoNext {
    DEVI void operator()() const {}
};
struct FwdFirst {
    Set4 w;
    f32x4 bias[OG][2];
};
DEVI FwdFirst load_fwd_first(const float *img, const float *tail, int lane, int q) {
    FwdFirst o;
    o.w = load_set(img, lane);
#pragma unroll
    for (int og = 0; og < OG; ++og) load_bias(o.bias[og], tail + og * 32, 0, q);
    return o;
}
// out[og] = act(bias + sum_ig W(og, ig) in[ig]); img = [OG][KGI] sets
template <int KGI, bool RELU, class Next>
DEVI void layer_fwd(float (&out)[OG][8], const float *img, const FwdFirst &first, const float (&in)[KGI][8], int lane, Next next) {
    constexpr int NS = OG * KGI;
    Set4 W[2];
    W[0] = first.w;
    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int og = i / KGI, ig = i % KGI;
        if (i + 1 < NS) W[(i + 1) & 1] = load_set(img + (i + 1) * SUB, lane);
        else next();
        __builtin_amdgcn_sched_barrier(0);           // (left alone the scheduler sinks the reads back to their first use)
        if (ig == 0) {
            acc[0] = first.bias[og][0];
            acc[1] = first.bias[og][1];
        }
        mfma_set(acc, W[i & 1], in[ig]);
        if (ig == KGI - 1) {
#pragma unroll
            for (int s = 0; s < 8; ++s) out[og][s] = RELU ? relu1(acc[s >> 2][s & 3]) : acc[s >> 2][s & 3];
        }
    }
}
// din[rg] = sum_kg W^T(rg, kg) d[kg]; img = [RG][OG] sets
template <int RG, class Next>
DEVI void layer_bwd(float (&out)[RG][8], const float *img, const Set4 &first, const float (&d)[OG][8], int lane, Next next) {
    constexpr int NS = RG * OG;
    Set4 W[2];
    W[0] = first;
    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int rg = i / OG, kg = i % OG;
        if (i + 1 < NS) W[(i + 1) & 1] = load_set(img + (i + 1) * SUB, lane);
        else next();
        __builtin_amdgcn_sched_barrier(0);
        if (kg == 0) acc[0] = acc[1] = zero4();
        mfma_set(acc, W[i & 1], d[kg]);
        if (kg == OG - 1) {
#pragma unroll
            for (int s = 0; s < 8; ++s) out[rg][s] = acc[s >> 2][s & 3];
        }
    }
}

// Weight gradients of one layer from the staged tiles: dW[og][ig][2 mb + nb] += Dt_og (rows 16 mb ..) x In_ig (rows 16 nb ..) over the
// 16 pixels (fgnn_t16.h wgrad16), db[og][mb] += pixel sums of Dt_og's rows (lane (i, q): pixels 4 q .. 4 q + 3 of row 16 mb + i).
// NI input tiles at TI, the accumulators of input group i0 + ii.
struct WgFirst {
    Pair4 a, b;
};
DEVI WgFirst load_wg_first(const float *TD, const float *TI, int lane) {
    WgFirst o;
    o.a = load_rows(TD, lane);
    o.b = load_rows(TI, lane);
    return o;
}
template <int KGT, int NI, bool BIAS, class Next>
DEVI void layer_wgrad(f32x4 (&dW)[OG][KGT][4], float (&db)[OG][2], const float *TD, const float *TI, int i0, const WgFirst &first, int lane,
                      Next next) {
    constexpr int NS = OG * NI;
    Pair4 A[2], B[2];
    A[0] = first.a;
    B[0] = first.b;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int og = i / NI, ii = i % NI;
        if (i + 1 < NS) {
            const int ogn = (i + 1) / NI, iin = (i + 1) % NI;
            if (iin == 0) A[ogn & 1] = load_rows(TD + ogn * TILE_F, lane);
            B[(i + 1) & 1] = load_rows(TI + iin * TILE_F, lane);
        } else {
            next();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (BIAS && ii == 0) {
            const Pair4 &a = A[og & 1];
            db[og][0] += (a.lo.x + a.lo.y) + (a.lo.z + a.lo.w);
            db[og][1] += (a.hi.x + a.hi.y) + (a.hi.z + a.hi.w);
        }
        wgrad_mfma(dW[og][i0 + ii], A[og & 1], B[i & 1]);
    }
}

// one tile's position: graph, per-lane pixel, load / store offsets (fragment lane (px, q))
struct Tile64 {
    int g;
    int p;
    bool inb, valid;
};
DEVI Tile64 tile64(int t, int total, int hpg, int P2, int N, float rcpN, const rsrc_t &rnv, bool ragged, int px) {
    Tile64 c;
    const bool act = t < total;
    c.g = __builtin_amdgcn_readfirstlane(act ? t / hpg : 0);
    c.p = (act ? t - c.g * hpg : 0) * 16 + px;
    c.inb = act && c.p < P2;
    c.valid = c.inb;
    if (ragged) {
        const int nv = graph_nv(rnv, true, c.g, N);
        int i, jj;
        row_col(c.p, N, rcpN, i, jj);
        c.valid = c.inb && i < nv && jj < nv;
    }
    return c;
}
// The input of the MLP: one slab, or two stacked along the channels (a block's mlp3 reads [mult ; in], models/blocks_emb.py:33-36, without
// the concatenated copy): groups 0 .. kga-1 are slab a's (a whole number of 32-channel groups when there is a slab b), the rest slab b's.
struct In2 {
    View a, b;
    int cin;            // channels of both
};
DEVI In2 make_in2(const float *pa, long long gsa, long long lda, int ca, const float *pb, long long gsb, long long ldb, int cb, int G) {
    In2 o;
    o.a = make_view(pa, gsa, lda, G);
    o.b = make_view(pb ? pb : pa, pb ? gsb : gsa, pb ? ldb : lda, G);
    o.cin = ca + (pb ? cb : 0);
    return o;
}
DEVI In2 make_in1(const float *p, long long gs, long long ld, int ch, int G) { return make_in2(p, gs, ld, ch, nullptr, 0, 0, 0, G); }
// 8 KG registers of the input; the channels >= cin of a partial last group are not touched (read as zeros)
template <int KG, int KGA = KG>        // KGA: 32-channel groups of slab a
DEVI void load_in(float (&x)[KG][8], const In2 &in, const Tile64 &c, int q) {
#pragma unroll
    for (int ig = 0; ig < KG; ++ig) {
        const bool ina = ig < KGA;
        const rsrc_t r = ina ? in.a.r : in.b.r;
        const int ld4 = ina ? in.a.ld4 : in.b.ld4, gs4 = ina ? in.a.gs4 : in.b.gs4, lg = ina ? ig : ig - KGA;
        const int voff = c.valid ? chan_q(q) * ld4 + 4 * c.p : OOB_OFF;
        const int s0 = c.g * gs4 + lg * 32 * ld4;
        if (ig == KG - 1 && in.cin < 32 * KG) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                x[ig][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 32 * ig + chan_s(s) + chan_q(q) < in.cin ? voff : OOB_OFF,
                                                                                          s0 + chan_s(s) * ld4, 0));
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) x[ig][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, s0 + chan_s(s) * ld4, 0));
        }
    }
}
// acc_a / acc_b: add to what the slab holds (one buffer_atomic_add_f32 per element: every element is touched by exactly one lane of
// one launch, so the sum old + new has one order) instead of storing
template <int KG, int KGA = KG>
DEVI void store_in(const float (&x)[KG][8], const In2 &in, const Tile64 &c, int q, bool acc_a, bool acc_b) {
#pragma unroll
    for (int ig = 0; ig < KG; ++ig) {
        const bool ina = ig < KGA;
        const rsrc_t r = ina ? in.a.r : in.b.r;
        const int ld4 = ina ? in.a.ld4 : in.b.ld4, gs4 = ina ? in.a.gs4 : in.b.gs4, lg = ina ? ig : ig - KGA;
        const int voff = c.inb ? chan_q(q) * ld4 + 4 * c.p : OOB_OFF;
        const int s0 = c.g * gs4 + lg * 32 * ld4;
        if (ina ? acc_a : acc_b) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bool ok = !(ig == KG - 1 && in.cin < 32 * KG) || 32 * ig + chan_s(s) + chan_q(q) < in.cin;
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(x[ig][s], r, ok ? voff : OOB_OFF, s0 + chan_s(s) * ld4, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bool ok = !(ig == KG - 1 && in.cin < 32 * KG) || 32 * ig + chan_s(s) + chan_q(q) < in.cin;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[ig][s]), r, ok ? voff : OOB_OFF, s0 + chan_s(s) * ld4, FGNN_STORE_AUX);
            }
        }
    }
}

// zeros for the existing pixels of a tile, slab by slab (do_a / do_b: the slab is being stored, not accumulated into)
template <int KG, int KGA = KG>
DEVI void zero_in(const In2 &in, const Tile64 &c, int q, bool do_a, bool do_b) {
#pragma unroll
    for (int ig = 0; ig < KG; ++ig) {
        const bool ina = ig < KGA;
        if (!(ina ? do_a : do_b)) continue;
        const rsrc_t r = ina ? in.a.r : in.b.r;
        const int ld4 = ina ? in.a.ld4 : in.b.ld4, gs4 = ina ? in.a.gs4 : in.b.gs4, lg = ina ? ig : ig - KGA;
        const int voff = c.inb ? chan_q(q) * ld4 + 4 * c.p : OOB_OFF;
        const int s0 = c.g * gs4 + lg * 32 * ld4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool ok = !(ig == KG - 1 && in.cin < 32 * KG) || 32 * ig + chan_s(s) + chan_q(q) < in.cin;
            __builtin_amdgcn_raw_buffer_store_b32(0u, r, ok ? voff : OOB_OFF, s0 + chan_s(s) * ld4, FGNN_STORE_AUX);
        }
    }
}

template <int KG>
struct FwdLayout {
    static constexpr int F0 = 0, F1 = F0 + OG * KG * SUB, F2 = F1 + OG * OG * SUB, BIAS = F2 + OG * OG * SUB;
    static constexpr int LDS_F = BIAS + 3 * 64;
};

template <int KG, int KGA>
__global__ __launch_bounds__(64 * NWF) void mlp64_fwd_kernel(const fgnn_mlp64_args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = FwdLayout<KG>;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = lane & 15, q = lane >> 4;
    const int P2 = A.N * A.N, hpg = (P2 + 15) / 16, total = A.G * hpg;
    const In2 vx = make_in2(A.x, A.x_gstride, A.x_ld, A.cin, A.xb, A.xb_gstride, A.xb_ld, A.cb, A.G);
    const View vo = make_view(A.out, A.o_gstride, A.o_ld, A.G);
    const bool ragged = A.nvalid != nullptr;
    const rsrc_t rnv = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(A.nvalid), 0, ragged ? A.G * 4 : 0, 0x00020000);
    const float rcpN = 1.f / (float)A.N;
    const int step = gridDim.x * NWF;

    int t = blockIdx.x * NWF + wv;
    float x[KG][8];
    Tile64 c = tile64(t, total, hpg, P2, A.N, rcpN, rnv, ragged, px);
    load_in<KG, KGA>(x, vx, c, q);                         // in flight under the operand copy

    {
        constexpr Packed64 PL{KG};
        copy4(smem + L::F0, A.packed + PL.f0(), (PL.b2() - PL.f0()) / 4, threadIdx.x, 64 * NWF);                 // F0, F1
        copy4(smem + L::F2, A.packed + PL.f2(), (PL.floats() - PL.f2()) / 4, threadIdx.x, 64 * NWF);             // F2, bias
    }
    __syncthreads();

    FwdFirst f0 = load_fwd_first(smem + L::F0, smem + L::BIAS, lane, q);
    while (t < total) {
        const Tile64 cn = tile64(t + step, total, hpg, P2, A.N, rcpN, rnv, ragged, px);
        float h1[OG][8], h2[OG][8], z[OG][8];
        asm volatile("" ::: "memory");                     // the operand reads are loop invariant: keep them from being hoisted into registers
        const bool live = !ragged || __ballot(c.valid) != 0ull;     // a tile without a valid pixel (ragged batches): zeros, no arithmetic
        if (live) {
            FwdFirst f1, f2;
            layer_fwd<KG, true>(h1, smem + L::F0, f0, x, lane, [&] { f1 = load_fwd_first(smem + L::F1, smem + L::BIAS + 64, lane, q); });
            load_in<KG, KGA>(x, vx, cn, q);                // the next tile into the registers just consumed
            layer_fwd<OG, true>(h2, smem + L::F1, f1, h1, lane, [&] { f2 = load_fwd_first(smem + L::F2, smem + L::BIAS + 128, lane, q); });
            layer_fwd<OG, false>(z, smem + L::F2, f2, h2, lane, [&] { f0 = load_fwd_first(smem + L::F0, smem + L::BIAS, lane, q); });
        } else {
            load_in<KG, KGA>(x, vx, cn, q);
#pragma unroll
            for (int og = 0; og < OG; ++og)
#pragma unroll
                for (int s = 0; s < 8; ++s) z[og][s] = 0.f;
        }
        const int voff = lane_voff(vo, q, c.p, c.inb);
#pragma unroll
        for (int og = 0; og < OG; ++og) {
#pragma unroll
            for (int s = 0; s < 8; ++s) z[og][s] = c.valid ? z[og][s] : 0.f;
            store8(z[og], vo, voff, c.g * vo.gs4 + og * 32 * vo.ld4);
        }
        c = cn;
        t += step;
    }
}

template <int KG, bool HAS_DX>
struct BwdLayout {
    static constexpr int F0 = 0, F1 = F0 + OG * KG * SUB;                                   // recompute: layers 0, 1
    static constexpr int B2 = F1 + OG * OG * SUB, B1 = B2 + OG * OG * SUB, B0 = B1 + OG * OG * SUB;
    static constexpr int BIAS = B0 + (HAS_DX ? KG * OG * SUB : 0);
    static constexpr int NSLOT = 4;                                                         // per wave: two gradient tiles, two input tiles
    static constexpr int TILE_OFF = BIAS + 2 * 64;
    static constexpr int MAIN_F = TILE_OFF + NWB * NSLOT * TILE_F;
    static constexpr int K0P = 32 * KG;                                                     // padded input width of the partial rows
    static constexpr int PCOUNT = 64 * K0P + 64 + 2 * (64 * 64 + 64);
    static constexpr int RED_F = NWB * (64 * K0P + 64);                                     // the epilogue's four records of the widest layer
    static constexpr int LDS_F = MAIN_F > RED_F ? MAIN_F : RED_F;
};

template <int KG, int KGA, bool HAS_DX>
__global__ __launch_bounds__(64 * NWB) void mlp64_bwd_kernel(const fgnn_mlp64_args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using L = BwdLayout<KG, HAS_DX>;
    constexpr int IH = KG > 2 ? 2 : KG;                   // input tiles staged at once (K0 = 128: two halves)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = lane & 15, q = lane >> 4;
    const int P2 = A.N * A.N, hpg = (P2 + 15) / 16, total = A.G * hpg;
    const In2 vx = make_in2(A.x, A.x_gstride, A.x_ld, A.cin, A.xb, A.xb_gstride, A.xb_ld, A.cb, A.G);
    const In2 vdz = make_in1(A.dz, A.dz_gstride, A.dz_ld, 64, A.G);
    const In2 vdx = make_in2(A.dx, A.dx_gstride, A.dx_ld, A.cin, A.xb ? A.dxb : nullptr, A.dxb_gstride, A.dxb_ld, A.cb, A.G);
    const bool ragged = A.nvalid != nullptr;
    const rsrc_t rnv = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(A.nvalid), 0, ragged ? A.G * 4 : 0, 0x00020000);
    const float rcpN = 1.f / (float)A.N;
    const int step = gridDim.x * NWB;
    float *my = smem + L::TILE_OFF + wv * (L::NSLOT * TILE_F);
    float *TD = my, *TI = my + 2 * TILE_F;
    const int lane_base = tile_lane_base(px, q);

    f32x4 dW0[OG][KG][4], dW1[OG][OG][4], dW2[OG][OG][4];
    float db0[OG][2], db1[OG][2], db2[OG][2];
#pragma unroll
    for (int og = 0; og < OG; ++og) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int ig = 0; ig < KG; ++ig) dW0[og][ig][k] = zero4();
#pragma unroll
            for (int ig = 0; ig < OG; ++ig) dW1[og][ig][k] = dW2[og][ig][k] = zero4();
        }
        db0[og][0] = db0[og][1] = db1[og][0] = db1[og][1] = db2[og][0] = db2[og][1] = 0.f;
    }

    // the wave's next tile with a valid pixel after `tcur`; the padding-only tiles in between (ragged batches) get their zero dx here -- not
    // by a branch around the tile body: accumulators that are live across a branch cost a second register set (36 - 190 spilled registers)
    auto advance = [&](int tcur) {
        int tn = tcur + step;
        if (ragged) {
            while (tn < total) {
                const Tile64 cc = tile64(tn, total, hpg, P2, A.N, rcpN, rnv, true, px);
                if (__ballot(cc.valid) != 0ull) break;
                if constexpr (HAS_DX) zero_in<KG, KGA>(vdx, cc, q, A.accumulate_dx == 0, A.accumulate_dxb == 0);
                tn += step;
            }
        }
        return __builtin_amdgcn_readfirstlane(tn);
    };
    int t = advance(blockIdx.x * NWB + wv - step);
    float x[KG][8], dz[OG][8];
    Tile64 c = tile64(t, total, hpg, P2, A.N, rcpN, rnv, ragged, px);
    load_in<KG, KGA>(x, vx, c, q);
    load_in<OG>(dz, vdz, c, q);

#ifndef M64_ABL
#define M64_ABL 0           // measurement switch (tools/build_variant.sh): 1 no operand copy, 2 no parameter-gradient epilogue
#endif
    if (!(M64_ABL & 1)) {
        constexpr Packed64 PL{KG};
        copy4(smem + L::F0, A.packed + PL.f0(), ((HAS_DX ? PL.f2() : PL.b0()) - PL.f0()) / 4, threadIdx.x, 64 * NWB);   // F0, F1, B2, B1 (, B0)
        copy4(smem + L::BIAS, A.packed + PL.bias(), 2 * 64 / 4, threadIdx.x, 64 * NWB);
    }
    __syncthreads();

    FwdFirst f0 = load_fwd_first(smem + L::F0, smem + L::BIAS, lane, q);
    while (t < total) {
        const int tnext = advance(t);
        const Tile64 cn = tile64(tnext, total, hpg, P2, A.N, rcpN, rnv, ragged, px);
        float xn[KG][8], dzn[OG][8];
        load_in<KG, KGA>(xn, vx, cn, q);                   // the next tile's operands fly during this tile
        load_in<OG>(dzn, vdz, cn, q);

        // ---- recompute (the forward's fma sequence) ----
        float h1[OG][8], h2[OG][8];
        FwdFirst f1;
        Set4 wB2, wB1, wB0;
        WgFirst g;
        layer_fwd<KG, true>(h1, smem + L::F0, f0, x, lane, [&] { f1 = load_fwd_first(smem + L::F1, smem + L::BIAS + 64, lane, q); });
        layer_fwd<OG, true>(h2, smem + L::F1, f1, h1, lane, [&] { wB2 = load_set(smem + L::B2, lane); });

        // ---- layer 2: d2 = dz (zero outside the valid corner: those lanes loaded nothing).  The input-gradient GEMM runs first: the
        //      weight-gradient GEMM's tile reads are then requested under its last MFMAs ----
        float d[OG][8];
#pragma unroll
        for (int og = 0; og < OG; ++og) {
            stage8(TD + og * TILE_F, lane_base, dz[og]);
            stage8(TI + og * TILE_F, lane_base, h2[og]);
        }
        layer_bwd<OG>(d, smem + L::B2, wB2, dz, lane, [&] { g = load_wg_first(TD, TI, lane); });
        layer_wgrad<OG, OG, true>(dW2, db2, TD, TI, 0, g, lane, [&] { wB1 = load_set(smem + L::B1, lane); });
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int s = 0; s < 8; ++s) d[og][s] = h2[og][s] > 0.f ? d[og][s] : 0.f;

        // ---- layer 1 ----
#pragma unroll
        for (int og = 0; og < OG; ++og) {
            stage8(TD + og * TILE_F, lane_base, d[og]);
            stage8(TI + og * TILE_F, lane_base, h1[og]);
        }
        float d0[OG][8];
        layer_bwd<OG>(d0, smem + L::B1, wB1, d, lane, [&] { g = load_wg_first(TD, TI, lane); });
        layer_wgrad<OG, OG, true>(dW1, db1, TD, TI, 0, g, lane, [&] {
            if constexpr (HAS_DX) wB0 = load_set(smem + L::B0, lane);
        });
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int s = 0; s < 8; ++s) d0[og][s] = h1[og][s] > 0.f ? d0[og][s] : 0.f;

        // ---- layer 0 ----
#pragma unroll
        for (int og = 0; og < OG; ++og) stage8(TD + og * TILE_F, lane_base, d0[og]);
#pragma unroll
        for (int ii = 0; ii < IH; ++ii) stage8(TI + ii * TILE_F, lane_base, x[ii]);
        if constexpr (HAS_DX) {
            float dx[KG][8];
            // accumulating launches (the input also feeds other MLPs): up to two groups -- mlp1 / mlp2 of a block -- read what the buffer
            // holds while the GEMM runs and store the sum; wider inputs (no registers left) add with buffer atomics
            constexpr bool RMW = KG <= 2 && KGA == KG;
            float old[RMW ? KG : 1][8];
            const bool acc_a = A.accumulate_dx != 0;
            if constexpr (RMW) {
                if (acc_a) {
                    Tile64 cs = c;
                    cs.valid = c.inb;
                    load_in<KG, KGA>(old, vdx, cs, q);
                }
            }
            layer_bwd<KG>(dx, smem + L::B0, wB0, d0, lane, [&] { g = load_wg_first(TD, TI, lane); });
            if constexpr (RMW) {
                if (acc_a) {
#pragma unroll
                    for (int ig = 0; ig < KG; ++ig)
#pragma unroll
                        for (int s = 0; s < 8; ++s) dx[ig][s] += old[ig][s];
                }
                store_in<KG, KGA>(dx, vdx, c, q, false, false);
            } else {
                store_in<KG, KGA>(dx, vdx, c, q, acc_a, A.accumulate_dxb != 0);
            }
        } else {
            g = load_wg_first(TD, TI, lane);
        }
        if constexpr (KG > IH) {                            // input groups 2, 3 through the same two tiles
            layer_wgrad<KG, IH, true>(dW0, db0, TD, TI, 0, g, lane, NoNext{});
#pragma unroll
            for (int ii = 0; ii < KG - IH; ++ii) stage8(TI + ii * TILE_F, lane_base, x[IH + ii]);
            g = load_wg_first(TD, TI, lane);
            layer_wgrad<KG, KG - IH, false>(dW0, db0, TD, TI, IH, g, lane, [&] { f0 = load_fwd_first(smem + L::F0, smem + L::BIAS, lane, q); });
        } else {
            layer_wgrad<KG, IH, true>(dW0, db0, TD, TI, 0, g, lane, [&] { f0 = load_fwd_first(smem + L::F0, smem + L::BIAS, lane, q); });
        }
#pragma unroll
        for (int ig = 0; ig < KG; ++ig)
#pragma unroll
            for (int s = 0; s < 8; ++s) x[ig][s] = xn[ig][s];
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int s = 0; s < 8; ++s) dz[og][s] = dzn[og][s];
        c = cn;
        t = tnext;
    }

    // ---- workgroup sum of the parameter gradients: the waves add their fragments to one LDS record in turn (fixed order) ----
    // record: [W0 (64 x K0P) | b0 (64) | W1 (64 x 64) | b1 | W2 | b2]
#pragma unroll
    for (int og = 0; og < OG; ++og)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            db0[og][b] += __shfl_xor(db0[og][b], 16);
            db0[og][b] += __shfl_xor(db0[og][b], 32);
            db1[og][b] += __shfl_xor(db1[og][b], 16);
            db1[og][b] += __shfl_xor(db1[og][b], 32);
            db2[og][b] += __shfl_xor(db2[og][b], 16);
            db2[og][b] += __shfl_xor(db2[og][b], 32);
        }
    // one layer at a time: every wave drops its fragments into its own LDS record [W (64 x KP) | b (64)] (plain stores), then the
    // workgroup adds the four records in wave order and writes the row piece of wpart
    constexpr int K0P = L::K0P;
    float *outp = A.wpart + (long long)blockIdx.x * L::PCOUNT;
    auto reduce_layer = [&](auto &dW, float (&db)[OG][2], auto kgl_tag, int out_off) {
        constexpr int KGL = decltype(kgl_tag)::value, KP = 32 * KGL, REC = 64 * KP + 64;
        static_assert(NWB * REC <= L::LDS_F, "layer records fit the workgroup's LDS");
        __syncthreads();                    // everyone is done with what the records overwrite
        float *red = smem + wv * REC;
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = 32 * og + row_chan(16 * mb + 4 * q + r), cc = row_chan(16 * nb + px);
#pragma unroll
                        for (int ig = 0; ig < KGL; ++ig) red[o * KP + 32 * ig + cc] = dW[og][ig][2 * mb + nb][r];
                    }
        if (q == 0) {
#pragma unroll
            for (int og = 0; og < OG; ++og)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) red[64 * KP + 32 * og + row_chan(16 * mb + px)] = db[og][mb];
        }
        __syncthreads();
        const float4 *r4 = reinterpret_cast<const float4 *>(smem);
        float4 *o4 = reinterpret_cast<float4 *>(outp + out_off);
        for (int e = threadIdx.x; e < REC / 4; e += 64 * NWB) {
            float4 v = r4[e];
#pragma unroll
            for (int w = 1; w < NWB; ++w) {                                  // fixed order
                const float4 u = r4[w * (REC / 4) + e];
                v.x += u.x;
                v.y += u.y;
                v.z += u.z;
                v.w += u.w;
            }
            o4[e] = v;
        }
    };
    if (M64_ABL & 2) return;
    reduce_layer(dW0, db0, std::integral_constant<int, KG>{}, 0);
    reduce_layer(dW1, db1, std::integral_constant<int, OG>{}, 64 * K0P + 64);
    reduce_layer(dW2, db2, std::integral_constant<int, OG>{}, 64 * K0P + 64 + 64 * 64 + 64);
}

template <int KG, int KGA = KG>
int launch_fwd(const fgnn_mlp64_args *a, hipStream_t st) {
    constexpr int LDS = FwdLayout<KG>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp64_fwd_kernel<KG, KGA>, LDS);
    hipLaunchKernelGGL((mlp64_fwd_kernel<KG, KGA>), dim3(WG64), dim3(64 * NWF), LDS, st, *a);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int KG, int KGA, bool HAS_DX>
int launch_bwd(const fgnn_mlp64_args *a, hipStream_t st) {
    constexpr int LDS = BwdLayout<KG, HAS_DX>::LDS_F * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)mlp64_bwd_kernel<KG, KGA, HAS_DX>, LDS);
    hipLaunchKernelGGL((mlp64_bwd_kernel<KG, KGA, HAS_DX>), dim3(WG64), dim3(64 * NWB), LDS, st, *a);
    FGNN_LAUNCH_CHECK();
    return 0;
}

int check_common(const fgnn_mlp64_args *a, const char *who) {
    FGNN_CHECK(a != nullptr, "%s: null args", who);
    const int cin = a->cin + (a->xb ? a->cb : 0);
    FGNN_CHECK(fgnn_mlp64_supported(cin, 3, 64), "%s: built for depth 3, 64 hidden / output channels and 1..128 input channels (got %d)", who, cin);
    FGNN_CHECK(!a->xb || (a->cin == 64 && a->cb > 0), "%s: a second slab is built for a 64-channel first slab (got %d)", who, a->cin);
    FGNN_CHECK(!a->xb || (long long)a->G * a->xb_gstride < 0x7fffffffll / 4, "%s: xb exceeds 2 GiB (32-bit buffer addressing); split the batch", who);
    FGNN_CHECK(a->G > 0 && a->N > 0 && a->N <= 256 && a->x, "%s: bad G / N (<= 256) / x", who);
    FGNN_CHECK(a->packed != nullptr, "%s: missing packed operand record (fgnn_mlp64_pack)", who);
    const long long lim = 0x7fffffffll / 4, G = a->G;
    FGNN_CHECK(G * a->x_gstride < lim, "%s: x exceeds 2 GiB (32-bit buffer addressing); split the batch", who);
    FGNN_CHECK((long long)a->G * ((a->N * a->N + 15) / 16) < (1ll << 29), "%s: too many tiles", who);
    return 0;
}

}  // namespace

extern "C" int fgnn_mlp64_supported(int cin, int depth, int width) {
    return depth == 3 && width == 64 && cin >= 1 && cin <= 128 ? 1 : 0;
}
extern "C" int fgnn_mlp64_num_workgroups(void) { return WG64; }
// floats per row of wpart: [W0 (64 x K0P) | b0 (64) | W1 (64 x 64) | b1 (64) | W2 (64 x 64) | b2 (64)], K0P = cin rounded up to 32
extern "C" int fgnn_mlp64_param_count(int cin) {
    const int k0p = (cin + 31) / 32 * 32;
    return 64 * k0p + 64 + 2 * (64 * 64 + 64);
}

extern "C" int fgnn_mlp64_packed_floats(int cin) { return packed_floats((cin + 31) / 32); }

// the operand records of up to 16 MLPs from their nn.Conv2d parameters (W0 (64, cin), W1, W2 (64, 64) row-major, biases (64) or NULL) in ONE launch
extern "C" int fgnn_mlp64_pack_multi(const fgnn_mlp64_pack_job *jobs, int njobs, void *stream) {
    FGNN_CHECK(jobs && njobs >= 1 && njobs <= PACK_MAX_JOBS, "fgnn_mlp64_pack_multi: 1..%d jobs", PACK_MAX_JOBS);
    PackJobs J = {};
    int most = 0;
    for (int j = 0; j < njobs; ++j) {
        const fgnn_mlp64_pack_job &b = jobs[j];
        FGNN_CHECK(fgnn_mlp64_supported(b.cin, 3, 64), "fgnn_mlp64_pack_multi: job %d: 1..128 input channels (got %d)", j, b.cin);
        FGNN_CHECK(b.W[0] && b.W[1] && b.W[2] && b.packed, "fgnn_mlp64_pack_multi: job %d: missing weights / output", j);
        J.W0[j] = b.W[0]; J.W1[j] = b.W[1]; J.W2[j] = b.W[2];
        J.b0[j] = b.bias[0]; J.b1[j] = b.bias[1]; J.b2[j] = b.bias[2];
        J.out[j] = b.packed;
        J.cin[j] = b.cin;
        const int total = packed_floats((b.cin + 31) / 32);
        most = total > most ? total : most;
    }
    hipLaunchKernelGGL(mlp64_pack_kernel, dim3((most + 1023) / 1024, njobs), dim3(256), 0, (hipStream_t)stream, J);
    FGNN_LAUNCH_CHECK();
    return 0;
}
extern "C" int fgnn_mlp64_pack(const float *W0, const float *W1, const float *W2, const float *b0, const float *b1, const float *b2, int cin,
                               float *packed, void *stream) {
    const fgnn_mlp64_pack_job job = {{W0, W1, W2}, {b0, b1, b2}, cin, packed};
    return fgnn_mlp64_pack_multi(&job, 1, stream);
}

extern "C" int fgnn_mlp64_fwd(const fgnn_mlp64_args *a, void *stream) {
    if (int rc = check_common(a, "fgnn_mlp64_fwd")) return rc;
    FGNN_CHECK(a->out != nullptr, "fgnn_mlp64_fwd: missing out");
    FGNN_CHECK((long long)a->G * a->o_gstride < 0x7fffffffll / 4, "fgnn_mlp64_fwd: out exceeds 2 GiB (32-bit buffer addressing); split the batch");
    hipStream_t st = (hipStream_t)stream;
    const int cin = a->cin + (a->xb ? a->cb : 0);
    if (a->xb) return cin <= 96 ? launch_fwd<3, 2>(a, st) : launch_fwd<4, 2>(a, st);
    if (cin <= 32) return launch_fwd<1>(a, st);
    if (cin <= 64) return launch_fwd<2>(a, st);
    if (cin <= 96) return launch_fwd<3>(a, st);
    return launch_fwd<4>(a, st);
}

extern "C" int fgnn_mlp64_bwd(const fgnn_mlp64_args *a, void *stream) {
    if (int rc = check_common(a, "fgnn_mlp64_bwd")) return rc;
    FGNN_CHECK(a->dz && a->wpart, "fgnn_mlp64_bwd: missing dz / wpart");
    const long long lim = 0x7fffffffll / 4, G = a->G;
    FGNN_CHECK(G * a->dz_gstride < lim && (!a->dx || G * a->dx_gstride < lim), "fgnn_mlp64_bwd: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    hipStream_t st = (hipStream_t)stream;
    const bool dx = a->dx != nullptr;
    FGNN_CHECK(!dx || !a->xb || (a->dxb && G * a->dxb_gstride < lim), "fgnn_mlp64_bwd: dx of a two-slab input needs dxb (< 2 GiB)");
    const int cin = a->cin + (a->xb ? a->cb : 0);
    if (a->xb) {
        if (cin <= 96) return dx ? launch_bwd<3, 2, true>(a, st) : launch_bwd<3, 2, false>(a, st);
        return dx ? launch_bwd<4, 2, true>(a, st) : launch_bwd<4, 2, false>(a, st);
    }
    if (cin <= 32) return dx ? launch_bwd<1, 1, true>(a, st) : launch_bwd<1, 1, false>(a, st);
    if (cin <= 64) return dx ? launch_bwd<2, 2, true>(a, st) : launch_bwd<2, 2, false>(a, st);
    if (cin <= 96) return dx ? launch_bwd<3, 3, true>(a, st) : launch_bwd<3, 3, false>(a, st);
    return dx ? launch_bwd<4, 4, true>(a, st) : launch_bwd<4, 4, false>(a, st);
}
