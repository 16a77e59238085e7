// Per-channel N x N matrix products on bf16 slabs (Matmul, models/layers.py:161-162) and their backward for gfx950,
// N <= 256.  One 512-thread workgroup owns one (g,c) matrix: every operand element is read from HBM once (16-byte loads),
// normalised on load ((z - mean) a + beta in fp32, padding -> 0), rounded to bf16 and streamed through double-buffered LDS
// panels in 64-wide k chunks; all ceil(nv/32)^2 output tiles of 32x32 live in the fp32 accumulators of the eight waves and are
// multiplied with v_mfma_f32_32x32x16_bf16.
//   Out[m][n] = sum_k OpA(m,k) OpB(k,n)
//   forward M = Ya Yb;   dA = dM Yb^T;   dB = Ya^T dM
// An operand reaches LDS as a plain copy of 16-byte row pieces of its source -- never transposed by the staging code:
//   * a source whose rows are the operand's NON-contracted index (Ya in the forward, dM and Yb in dA) fills an [x][k] panel
//     (68-element rows); an MFMA operand of one k-step is two conflict-free ds_read_b64;
//   * a source whose rows are the CONTRACTED index (Yb in the forward, Ya and dM in dB) fills a [k][x] panel; the MFMA operand
//     (lane = x, eight consecutive k) is fetched with ds_read_b64_tr_b16, the gfx950 LDS transpose read: within 16 lanes, lane i
//     supplies the address of row i>>2, 8-byte piece i&3 of a 4 x 16 block and lane l receives column l of it
//     (semantics measured with tools/ubench_tr16.hip).
// Matrix rows are `ldr` elements apart (ldr % 8 == 0), so every 16-byte access is aligned.
// N > 128: wave w owns the tile ROW w (its A operand of a k-step is read once and reused for every tile column).
#include <type_traits>
#include "fgnn_bf16.h"
#include "fgnn_norm.h"

#ifndef MM_ABLATE
#define MM_ABLATE 0          // debug builds (tools/gpu_mm16_ablate.py): 1 no stores, 2 no MFMA, 3 no LDS staging, 4 no global loads,
#endif                       // 5 phase time stamps of the forward kernel (fgnn_debug_mm16_stamps)

#if MM_ABLATE == 5
__device__ unsigned long long mm_stamps[1024][16];
#define MM_STAMP(k)                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 1024) mm_stamps[blockIdx.x][k] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define MM_STAMP(k) do {} while (0)
#endif

namespace {

constexpr int MM_NW = 8, MM_THREADS = 64 * MM_NW, MM_KC = 64;
constexpr int XK_LD = 136;                 // bytes per row of an [x][k] panel (64 k's + 4 elements of padding)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Src16 {
    View16 v;
    int off2;          // byte offset of the (g,c) matrix inside the view
    float a, b;        // y = x * a + b  (b = beta - mean * a); raw slabs: norm == false
    float mean;
    bool norm;
};

template <int NT, int NCOL = NT>
struct MMCfg {
    static constexpr int XM = 32 * NT;
    static constexpr int KX_LD = XM * 2 + 64;                       // bytes per row of a [k][x] panel: (KX_LD / 4) % 64 == 16, so
                                                                    // the 4 rows x 64 bytes of a transpose read tile all 64 banks
    static constexpr int XK_BYTES = XM * XK_LD, KX_BYTES = MM_KC * KX_LD;
    static constexpr int PANEL_B = XK_BYTES > KX_BYTES ? XK_BYTES : KX_BYTES;
    static constexpr int BUF_B = 2 * PANEL_B;                       // A + B panel
    static constexpr int LDS_BYTES = 2 * BUF_B;                     // double buffered
    // tile ownership: NT == 8 (N > 128): wave w owns the tile ROW w; smaller matrices keep the cyclic assignment
    static constexpr bool STRIP = NT == 8;
    static constexpr int MAXT = STRIP ? NCOL : (NT * NT + MM_NW - 1) / MM_NW;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};
template <int NT, int NCOL>
using AccArray = f32x16[MMCfg<NT, NCOL>::MAXT];

DEVI Src16 mm_src(const fgnn_slab16 &s, int G, int g, int c) {
    Src16 o;
    o.v = make_view16(s.ptr, s.gstride, s.ldp, G);
    o.off2 = g * o.v.gs2 + c * o.v.ld2;
    o.norm = s.nrm != nullptr;
    o.a = 1.f;
    o.b = 0.f;
    o.mean = 0.f;
    if (o.norm) {
        const float4 n = reinterpret_cast<const float4 *>(s.nrm)[(long long)g * s.C + c];
        const float be = s.beta ? s.beta[c] : 0.f;
        // uniform values: keep them in scalar registers
        auto sgpr = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
        o.a = sgpr(n.y);
        o.b = sgpr(be - n.x * n.y);
        o.mean = sgpr(n.x);
    }
    return o;
}
DEVI Src16 mm_src_plain(const void *p, long long gs, long long ld, int G, int g, int c) {
    Src16 o;
    o.v = make_view16(p, gs, ld, G);
    o.off2 = g * o.v.gs2 + c * o.v.ld2;
    o.norm = false;
    o.a = 1.f;
    o.b = 0.f;
    o.mean = 0.f;
    return o;
}

// eight consecutive elements of a row (row `r`, first element `e0`): normalise, zero the padding, round
DEVI u32x4 norm8(const u32x4 x, const Src16 &s, int nv, int r, int e0) {
    u32x4 o;
    if ((nv & 7) == 0) {                 // the piece is valid or padding as a whole: the mask folds into the affine pair
        const bool ok = r < nv && e0 < nv;
        const float a = ok ? s.a : 0.f, b = ok ? s.b : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = cvt_pk(fmaf(bf_lo(x[q]), a, b), fmaf(bf_hi(x[q]), a, b));
    } else {
        const float rm = r < nv ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float m0 = (e0 + 2 * q) < nv ? rm : 0.f, m1 = (e0 + 2 * q + 1) < nv ? rm : 0.f;
            o[q] = cvt_pk(fmaf(bf_lo(x[q]), s.a, s.b) * m0, fmaf(bf_hi(x[q]), s.a, s.b) * m1);
        }
    }
    return o;
}

// The same arithmetic one dword (two elements) at a time, for the pipelined kernels that slot the staging of the NEXT
// k chunk between the MFMAs of the current one.  The piece's validity folds into the affine pair; when nv is not a multiple
// of 8 the one piece per row that straddles nv is finished with an AND mask on the packed result.
struct PieceCtx {
    float a, b;
    unsigned mk[4];
};
template <bool WHOLE>
DEVI PieceCtx piece_ctx(const Src16 &s, int nv, int row, int e0) {
    PieceCtx c;
    const bool ok = row < nv && e0 < nv;
    c.a = ok ? s.a : 0.f;
    c.b = ok ? s.b : 0.f;
    if (!WHOLE) {
        const int lim = nv - e0;
#pragma unroll
        for (int q = 0; q < 4; ++q) c.mk[q] = lim >= 2 * q + 2 ? 0xffffffffu : (lim == 2 * q + 1 ? 0x0000ffffu : 0u);
    }
    return c;
}
template <bool WHOLE>
DEVI unsigned norm_dword(unsigned x, const PieceCtx &c, int q) {
    const unsigned o = cvt_pk(fmaf(bf_lo(x), c.a, c.b), fmaf(bf_hi(x), c.a, c.b));
    return WHOLE ? o : (o & c.mk[q]);
}

// ---- rows = non-contracted index -> [x][k] panel: thread -> (x = tid/8 + 64*sweep, 8 k's at k0 + 8*(tid%8)) --------------
template <int NT>
struct StageXK {
    static constexpr int SW = MMCfg<NT>::XM / 64;
    u32x4 r[SW];
    DEVI void load(const Src16 &s, int ldr, int nv, int k0, int tid) {
        const int pc = tid & 7, x = tid >> 3;
        const int kk = k0 + 8 * pc;
        const int base = kk < nv ? (x * ldr + kk) * 2 : OOB_OFF;
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            if (MM_ABLATE == 4) {
                r[i] = u32x4{(unsigned)tid, (unsigned)k0, 1u, 2u};
                continue;
            }
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(s.v.r, (x + 64 * i) < nv ? base : OOB_OFF, s.off2 + 64 * i * ldr * 2, 0);
        }
    }
    DEVI void stage(char *P, const Src16 &s, int nv, int k0, int tid) const {
        const int pc = tid & 7, x = tid >> 3;
        char *dst = P + x * XK_LD + pc * 16;
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            if (MM_ABLATE == 3) {
                asm volatile("" ::"v"(r[i][0]), "v"(r[i][1]), "v"(r[i][2]), "v"(r[i][3]));
                continue;
            }
            const u32x4 o = s.norm ? norm8(r[i], s, nv, x + 64 * i, k0 + 8 * pc) : r[i];
            uint2 *d = reinterpret_cast<uint2 *>(dst + 64 * i * XK_LD);       // rows are 8-byte (not 16-byte) aligned
            d[0] = make_uint2(o[0], o[1]);
            d[1] = make_uint2(o[2], o[3]);
        }
    }
    static DEVI u32x4 load_piece(int i, const Src16 &s, int ldr, int nv, int k0, int tid) {
        const int pc = tid & 7, x = tid >> 3, kk = k0 + 8 * pc;
        if (MM_ABLATE == 4) return u32x4{(unsigned)tid, (unsigned)k0, 1u, 2u};
        return __builtin_amdgcn_raw_buffer_load_b128(s.v.r, (kk < nv && (x + 64 * i) < nv) ? (x * ldr + kk) * 2 : OOB_OFF,
                                                     s.off2 + 64 * i * ldr * 2, 0);
    }
    DEVI void load_one(int i, const Src16 &s, int ldr, int nv, int k0, int tid) { r[i] = load_piece(i, s, ldr, nv, k0, tid); }
    template <bool WHOLE>
    DEVI PieceCtx ctx(const Src16 &s, int nv, int k0, int i, int tid) const {
        return piece_ctx<WHOLE>(s, nv, (tid >> 3) + 64 * i, k0 + 8 * (tid & 7));
    }
    DEVI void write(char *P, int i, const u32x4 o, int tid) const {
        uint2 *d = reinterpret_cast<uint2 *>(P + ((tid >> 3) + 64 * i) * XK_LD + (tid & 7) * 16);
        d[0] = make_uint2(o[0], o[1]);
        d[1] = make_uint2(o[2], o[3]);
    }
};
// ---- rows = contracted index -> [k][x] panel: thread -> (k = tid/PPR + RPS*sweep, 8 x's at 8*(tid%PPR)) ------------------
template <int NT>
struct StageKX {
    static constexpr int PPR = MMCfg<NT>::XM / 8, RPS = MM_THREADS / PPR, SW = MM_KC / RPS;
    u32x4 r[SW];
    DEVI void load(const Src16 &s, int ldr, int nv, int k0, int tid) {
        const int pc = tid % PPR, kr = tid / PPR;
        // per-lane part in the vector offset, the sweep's row block in the (uniform) scalar offset
        const int base = 8 * pc < nv ? 16 * pc + kr * ldr * 2 : OOB_OFF;
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            if (MM_ABLATE == 4) {
                r[i] = u32x4{(unsigned)tid, (unsigned)k0, 1u, 2u};
                continue;
            }
            const int kb = k0 + RPS * i;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(s.v.r, kb + kr < nv ? base : OOB_OFF, s.off2 + kb * ldr * 2, 0);
        }
    }
    DEVI void stage(char *P, const Src16 &s, int nv, int k0, int tid) const {
        const int pc = tid % PPR, kr = tid / PPR;
        char *dst = P + kr * MMCfg<NT>::KX_LD + pc * 16;
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            if (MM_ABLATE == 3) {
                asm volatile("" ::"v"(r[i][0]), "v"(r[i][1]), "v"(r[i][2]), "v"(r[i][3]));
                continue;
            }
            const u32x4 o = s.norm ? norm8(r[i], s, nv, k0 + kr + RPS * i, 8 * pc) : r[i];
            *reinterpret_cast<u32x4 *>(dst + RPS * i * MMCfg<NT>::KX_LD) = o;
        }
    }
    static DEVI u32x4 load_piece(int i, const Src16 &s, int ldr, int nv, int k0, int tid) {
        const int pc = tid % PPR, kr = tid / PPR, kb = k0 + RPS * i;
        if (MM_ABLATE == 4) return u32x4{(unsigned)tid, (unsigned)k0, 1u, 2u};
        return __builtin_amdgcn_raw_buffer_load_b128(s.v.r, (8 * pc < nv && kb + kr < nv) ? 16 * pc + kr * ldr * 2 : OOB_OFF,
                                                     s.off2 + kb * ldr * 2, 0);
    }
    DEVI void load_one(int i, const Src16 &s, int ldr, int nv, int k0, int tid) { r[i] = load_piece(i, s, ldr, nv, k0, tid); }
    template <bool WHOLE>
    DEVI PieceCtx ctx(const Src16 &s, int nv, int k0, int i, int tid) const {
        return piece_ctx<WHOLE>(s, nv, k0 + tid / PPR + RPS * i, 8 * (tid % PPR));
    }
    DEVI void write(char *P, int i, const u32x4 o, int tid) const {
        *reinterpret_cast<u32x4 *>(P + (tid / PPR + RPS * i) * MMCfg<NT>::KX_LD + (tid % PPR) * 16) = o;
    }
};
template <int NT, bool XK>
struct Stage : std::conditional<XK, StageXK<NT>, StageKX<NT>>::type {};

// MFMA operand of k-step `step` for the 32 x-indices of strip t of a panel, lane = x
struct OperandAddr {
    int xk, kx;        // per-lane byte offsets inside an [x][k] / [k][x] panel
};
template <int NT>
DEVI OperandAddr operand_addr(int lane) {
    const int j = lane & 31, h = lane >> 5, i = lane & 15, cg = (lane >> 4) & 1;
    OperandAddr o;
    o.xk = j * XK_LD + 16 * h;
    o.kx = (8 * h + (i >> 2)) * MMCfg<NT>::KX_LD + (16 * cg + 4 * (i & 3)) * 2;
    return o;
}
template <int NT, bool XK>
DEVI i32x4 read_operand(const char *P, const OperandAddr &oa, int t, int step) {
    i32x4 o;
    if constexpr (XK) {
        const uint2 *p = reinterpret_cast<const uint2 *>(P + oa.xk + t * (32 * XK_LD) + step * 32);
        const uint2 a = p[0], b = p[1];
        o[0] = (int)a.x;
        o[1] = (int)a.y;
        o[2] = (int)b.x;
        o[3] = (int)b.y;
    } else {
        typedef __attribute__((address_space(3))) s16x4 *lds_p;
        constexpr int LD = MMCfg<NT>::KX_LD;
        const char *q = P + oa.kx + step * (16 * LD) + t * 64;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)q);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(q + 4 * LD));
        const uint2 a = __builtin_bit_cast(uint2, v0), b = __builtin_bit_cast(uint2, v1);
        o[0] = (int)a.x;
        o[1] = (int)a.y;
        o[2] = (int)b.x;
        o[3] = (int)b.y;
    }
    return o;
}

// ---- N > 128: software-pipelined version --------------------------------------------------------------------------------
// One register set holds the NEXT chunk.  While the MFMAs of chunk c run, the wave normalises chunk c+1 one dword per
// MFMA (the VALU work hides under the 8-pass matrix instructions of its own and of the SIMD's other wave), drops it into
// the other LDS buffer and refills the freed registers with the same piece of chunk c+2, which then has one whole chunk
// of MFMAs to arrive.  One k-step of the chunk = one staging sweep of each operand (MM_KC / 16 == sweeps == 4).
// Waves past the last tile row repeat the last one (their SIMD's matrix pipe would idle otherwise) and are skipped by
// mm_store; the last chunk stages the all-zero chunk past nv into the idle buffer.  Both keep the chunk one branch-free
// block with a single code path (a second path's join makes the register allocator ping-pong the accumulators).
// LAST: the product's final chunk stages nothing (the LDS buffers become the output image next); instead the registers
// receive chunk 0 of the NEXT product through `next(i, a_piece, b_piece)`, so its first loads fly during this chunk's MFMAs
// and the store epilogue.
struct NoNext {
    DEVI void operator()(int, u32x4 &, u32x4 &) const {}
};
template <int NT, int NCOL, bool A_XK, bool B_XK, bool A_PLAIN, bool B_PLAIN, bool WHOLE, bool LAST, class Next>
DEVI void strip_chunk(AccArray<NT, NCOL> &acc, const char *pa, const char *pb, const OperandAddr &oa, int strip,
                      Stage<NT, A_XK> &sa, Stage<NT, B_XK> &sb, char *nxa, char *nxb, const Src16 &A, const Src16 &B, int ldr,
                      int nv, int k0n, int tid, const Next &next) {
    if constexpr (LAST) {
#pragma unroll
        for (int i = 0; i < MM_KC / 16; ++i) next(i, sa.r[i], sb.r[i]);
    }
    static_assert(Stage<NT, A_XK>::SW == MM_KC / 16 && Stage<NT, B_XK>::SW == MM_KC / 16, "one sweep per k-step");
    static_assert(NCOL >= 4, "dword slots");
    // B operands run one MFMA ahead of their use; the order below is pinned with scheduling barriers (left alone the
    // scheduler clusters the MFMAs and spills the staging registers)
    constexpr int STEPS = MM_KC / 16, TOT = STEPS * NCOL;
    i32x4 bq[2];
    bq[0] = read_operand<NT, B_XK>(pb, oa, 0, 0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const i32x4 a = read_operand<NT, A_XK>(pa, oa, strip, s);
        PieceCtx cx;
#pragma unroll
        for (int tn = 0; tn < NCOL; ++tn) {
            const int it = s * NCOL + tn;
            if (it + 1 < TOT) bq[(it + 1) & 1] = read_operand<NT, B_XK>(pb, oa, (it + 1) % NCOL, (it + 1) / NCOL);
            if (MM_ABLATE != 2) acc[tn] = mfma16(bq[it & 1], a, acc[tn]);
            if (MM_ABLATE != 3 && !LAST) {
                if (tn < 4) {
                    if (!A_PLAIN) {
                        if (tn == 0) cx = sa.template ctx<WHOLE>(A, nv, k0n, s, tid);
                        sa.r[s][tn] = norm_dword<WHOLE>(sa.r[s][tn], cx, tn);
                    }
                    if (tn == 3) {
                        sa.write(nxa, s, sa.r[s], tid);
                        sa.load_one(s, A, ldr, nv, k0n + MM_KC, tid);
                    }
                } else if (!B_PLAIN) {
                    if (tn == 4) cx = sb.template ctx<WHOLE>(B, nv, k0n, s, tid);
                    sb.r[s][tn - 4] = norm_dword<WHOLE>(sb.r[s][tn - 4], cx, tn - 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MM_ABLATE != 3 && !LAST) {
            if (!B_PLAIN) {
#pragma unroll
                for (int q = NCOL - 4; q < 4; ++q) sb.r[s][q] = norm_dword<WHOLE>(sb.r[s][q], cx, q);
            }
            sb.write(nxb, s, sb.r[s], tid);
            sb.load_one(s, B, ldr, nv, k0n + MM_KC, tid);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// pre_a / pre_b: chunk 0 of this product when PRELOADED (left there by the previous product's last chunk); on return they
// hold whatever `next` loaded (chunk 0 of the following product).
template <int NT, int NCOL, bool A_XK, bool B_XK, bool A_PLAIN, bool B_PLAIN, bool WHOLE, bool PRELOADED, class Next>
DEVI void mm_gemm_strip(AccArray<NT, NCOL> &acc, const Src16 &A, const Src16 &B, char *lds, int ldr, int nv, int ntv, int tid,
                        u32x4 (&pre_a)[4], u32x4 (&pre_b)[4], const Next &next) {
    using Cf = MMCfg<NT, NCOL>;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wv < ntv ? wv : ntv - 1;
    const OperandAddr oa = operand_addr<NT>(lane);
#pragma unroll
    for (int ti = 0; ti < Cf::MAXT; ++ti) zero16f(acc[ti]);
    const int nkc = (nv + MM_KC - 1) / MM_KC;
    Stage<NT, A_XK> sa;
    Stage<NT, B_XK> sb;
    MM_STAMP(0);
    if constexpr (PRELOADED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa.r[i] = pre_a[i];
            sb.r[i] = pre_b[i];
        }
    } else {
        sa.load(A, ldr, nv, 0, tid);
        sb.load(B, ldr, nv, 0, tid);
    }
    MM_STAMP(1);
#pragma unroll
    for (int i = 0; i < MM_KC / 16; ++i) {                 // chunk 0: nothing to hide it under
        if (!A_PLAIN) {
            const PieceCtx c = sa.template ctx<WHOLE>(A, nv, 0, i, tid);
#pragma unroll
            for (int q = 0; q < 4; ++q) sa.r[i][q] = norm_dword<WHOLE>(sa.r[i][q], c, q);
        }
        sa.write(lds, i, sa.r[i], tid);
        sa.load_one(i, A, ldr, nv, MM_KC, tid);            // past nv: out-of-range offsets, no traffic
        if (!B_PLAIN) {
            const PieceCtx c = sb.template ctx<WHOLE>(B, nv, 0, i, tid);
#pragma unroll
            for (int q = 0; q < 4; ++q) sb.r[i][q] = norm_dword<WHOLE>(sb.r[i][q], c, q);
        }
        sb.write(lds + Cf::PANEL_B, i, sb.r[i], tid);
        sb.load_one(i, B, ldr, nv, MM_KC, tid);
    }
    __syncthreads();
    MM_STAMP(2);
    for (int c = 0; c + 1 < nkc; ++c) {
        const char *pa = lds + (c & 1) * Cf::BUF_B, *pb = pa + Cf::PANEL_B;
        char *nx = lds + ((c & 1) ^ 1) * Cf::BUF_B;
        strip_chunk<NT, NCOL, A_XK, B_XK, A_PLAIN, B_PLAIN, WHOLE, false>(acc, pa, pb, oa, strip, sa, sb, nx, nx + Cf::PANEL_B, A, B,
                                                                        ldr, nv, (c + 1) * MM_KC, tid, next);
        if (c < 4) MM_STAMP(3 + 2 * c);
        __syncthreads();
        if (c < 4) MM_STAMP(4 + 2 * c);
    }
    if (nkc > 0) {
        const int c = nkc - 1;
        const char *pa = lds + (c & 1) * Cf::BUF_B, *pb = pa + Cf::PANEL_B;
        strip_chunk<NT, NCOL, A_XK, B_XK, A_PLAIN, B_PLAIN, WHOLE, true>(acc, pa, pb, oa, strip, sa, sb, nullptr, nullptr, A, B, ldr, nv,
                                                                       0, tid, next);
        if (c < 4) MM_STAMP(3 + 2 * c);
        __syncthreads();
        if (c < 4) MM_STAMP(4 + 2 * c);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) next(i, sa.r[i], sb.r[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pre_a[i] = sa.r[i];
        pre_b[i] = sb.r[i];
    }
}

// acc (the wave's tiles of the ntv x ntv valid tiles) = (OpA OpB)^T over all k chunks: the MFMA takes OpB^T as its first and
// OpA^T as its second operand, so a lane ends up with one ROW of the product tile (what mm_store wants)
template <int NT, int NCOL, bool A_XK, bool B_XK, bool A_PLAIN = false, bool B_PLAIN = false, bool PRELOADED = false,
          class Next = NoNext>
DEVI void mm_gemm(AccArray<NT, NCOL> &acc, const Src16 &A, const Src16 &B, char *lds, int ldr, int nv, int ntv, int tid,
                  u32x4 (&pre_a)[4], u32x4 (&pre_b)[4], const Next &next = Next()) {
    using Cf = MMCfg<NT, NCOL>;
    if constexpr (Cf::STRIP) {
        if ((nv & 7) == 0)
            mm_gemm_strip<NT, NCOL, A_XK, B_XK, A_PLAIN, B_PLAIN, true, PRELOADED>(acc, A, B, lds, ldr, nv, ntv, tid, pre_a, pre_b, next);
        else
            mm_gemm_strip<NT, NCOL, A_XK, B_XK, A_PLAIN, B_PLAIN, false, PRELOADED>(acc, A, B, lds, ldr, nv, ntv, tid, pre_a, pre_b, next);
        return;
    }
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = ntv * ntv;
    const OperandAddr oa = operand_addr<NT>(lane);
#pragma unroll
    for (int ti = 0; ti < Cf::MAXT; ++ti) zero16f(acc[ti]);
    const int nkc = (nv + MM_KC - 1) / MM_KC;
    Stage<NT, A_XK> sa;
    Stage<NT, B_XK> sb;
    MM_STAMP(0);
    sa.load(A, ldr, nv, 0, tid);
    sb.load(B, ldr, nv, 0, tid);
    MM_STAMP(1);
    sa.stage(lds, A, nv, 0, tid);
    sb.stage(lds + Cf::PANEL_B, B, nv, 0, tid);
    __syncthreads();
    MM_STAMP(2);
    for (int c = 0; c < nkc; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nkc;
        if (more) {
            sa.load(A, ldr, nv, (c + 1) * MM_KC, tid);
            sb.load(B, ldr, nv, (c + 1) * MM_KC, tid);
        }
        const char *pa = lds + cur * Cf::BUF_B, *pb = pa + Cf::PANEL_B;
        if (MM_ABLATE != 2) {
            {
#pragma unroll
                for (int ti = 0; ti < Cf::MAXT; ++ti) {
                    const int t = wv + MM_NW * ti;
                    if (t < T) {
                        const int tm = t / ntv, tn = t - tm * ntv;
#pragma unroll
                        for (int s = 0; s < MM_KC / 16; ++s)
                            acc[ti] = mfma16(read_operand<NT, B_XK>(pb, oa, tn, s), read_operand<NT, A_XK>(pa, oa, tm, s), acc[ti]);
                    }
                }
            }
        }
        if (c < 4) MM_STAMP(3 + 2 * c);
        if (more) {
            char *nx = lds + (cur ^ 1) * Cf::BUF_B;
            sa.stage(nx, A, nv, (c + 1) * MM_KC, tid);
            sb.stage(nx + Cf::PANEL_B, B, nv, (c + 1) * MM_KC, tid);
        }
        __syncthreads();
        if (c < 4) MM_STAMP(4 + 2 * c);
    }
}

// accumulators -> global as bf16 through LDS.  The accumulators hold Out^T tiles (mm_gemm swaps the MFMA operands), so a lane
// owns one output ROW of a tile and four runs of four consecutive columns: it drops them into an [row][col] LDS image of the
// matrix with four ds_write_b64, and the workgroup then copies the image out in 16-byte row pieces (a matrix with ldr == N is
// one contiguous run).  Pieces outside the 32*ntv computed rows / columns are written as zeros, so the whole N x ldr output is
// defined.  Optionally S1 = sum t, S2 = sum t * (raw - mean) of the ROUNDED values over the valid entries, `raw` re-read from
// the un-normalised slab with the same 16-byte pieces.
template <int NT>
constexpr int out_pitch() { return MMCfg<NT>::XM * 2 + 16; }      // bytes; (pitch / 4) % 64 == 4: conflict-free 16-byte row reads

template <int NT, int NCOL, int STATS>
DEVI void mm_store(const AccArray<NT, NCOL> &acc, char *lds, const View16 &ov, int o_off2, const Src16 &raw, int N, int ldr,
                   int nv, int ntv, float &s1, float &s2, int tid) {
    constexpr int OP = out_pitch<NT>();
    static_assert(MMCfg<NT>::XM * OP <= MMCfg<NT>::LDS_BYTES, "output image must fit the staging panels");
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int T = ntv * ntv;
#pragma unroll
    for (int ti = 0; ti < MMCfg<NT, NCOL>::MAXT; ++ti) {
        const int t = MMCfg<NT, NCOL>::STRIP ? (wv < ntv && ti < ntv ? wv * ntv + ti : T) : wv + MM_NW * ti;
        if (t < T) {
            const int tm = MMCfg<NT, NCOL>::STRIP ? wv : t / ntv, tn = MMCfg<NT, NCOL>::STRIP ? ti : t - tm * ntv;
            char *p = lds + (32 * tm + j) * OP + (32 * tn + 4 * h) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q)                                       // columns 32 tn + 8 q + 4 h + {0..3}
                *reinterpret_cast<uint2 *>(p + 16 * q) =
                    make_uint2(cvt_pk(acc[ti][4 * q], acc[ti][4 * q + 1]), cvt_pk(acc[ti][4 * q + 2], acc[ti][4 * q + 3]));
        }
    }
    __syncthreads();
    MM_STAMP(11);
    // The next product's first chunk was requested during the last MFMA chunk.  Let it land BEFORE the stores below are
    // issued: vector memory operations of a wave retire in order, so waiting for those loads later (a runtime number of
    // stores in between) would mean waiting for every store to reach memory.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
    const int piece = tid & 31, rs = tid >> 5, X = 32 * ntv;
    if (8 * piece < ldr) {
        const bool col_in = 8 * piece < X;
        const bool whole = (nv & 7) == 0;
        const int cbase = piece * 16;
        const int voff = cbase + rs * ldr * 2;                                  // per-lane part; the row block is uniform
        const int rawc = (!whole || 8 * piece < nv) ? voff : OOB_OFF;
#pragma unroll 2
        for (int rb = 0; rb < N; rb += MM_THREADS / 32) {
            const int r = rb + rs;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (col_in && r < X) v = *reinterpret_cast<const u32x4 *>(lds + r * OP + cbase);
            u32x4 u;
            if (STATS == 1) u = __builtin_amdgcn_raw_buffer_load_b128(raw.v.r, r < nv ? rawc : OOB_OFF, raw.off2 + rb * ldr * 2, 0);
            if (MM_ABLATE == 1) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
            // soffset stays a literal 0: with an SGPR there the compiler assumes a VALU write to the data registers may follow
            // the 128-bit store immediately -- on gfx950 that corrupted dword 2 of some lanes (measured, tests/diag/gpu_mm16_check.py)
            else __builtin_amdgcn_raw_buffer_store_b128(v, ov.r, r < N ? voff + o_off2 + rb * ldr * 2 : OOB_OFF, 0, 0);
            if (STATS == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s1 += bf_lo(v[q]) + bf_hi(v[q]);
            }
            if (STATS == 1) {
                // outputs outside the valid nv x nv block are exact zeros (masked operands); `raw` there is replaced by 0 so
                // that stale padding bits cannot turn 0 * x into NaN
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x0 = bf_lo(u[q]), x1 = bf_hi(u[q]);
                    if (!whole) {
                        x0 = (8 * piece + 2 * q) < nv ? x0 : raw.mean;
                        x1 = (8 * piece + 2 * q + 1) < nv ? x1 : raw.mean;
                    }
                    const float t0 = bf_lo(v[q]), t1 = bf_hi(v[q]);
                    s1 += t0 + t1;
                    s2 = fmaf(t0, x0 - raw.mean, s2);
                    s2 = fmaf(t1, x1 - raw.mean, s2);
                }
            }
        }
    }
    MM_STAMP(12);
    __syncthreads();                      // the image is the next product's staging buffer
    MM_STAMP(13);
}

// GraphNorm finalize of the two operands folded into the forward product's prologue (the work of fgnn_gn_finalize2_tpg
// without its launch): the workgroup of matrix (g, c) reduces the tile statistics {mean_t, M2_t, n_t} of channel c of
// graph g for both operands -- the same exact two-level decomposition, two fixed-order workgroup sums -- while its first
// operand chunk is in flight, writes the records for the backward pass and uses them at once.
struct FinArgs16 {
    const float *part_a, *part_b, *cnt;      // (G, C, tpg, 2) {mean, M2} per tile and (G, tpg) valid elements per tile
    const float *gw_a, *gw_b;                // GraphNorm weights (C) or NULL = 1
    float *nrm_a, *nrm_b;                    // out: records (G*C*4)
    float eps;
    int tpg;
};
template <int K>
DEVI void wg_sum(float (&v)[K], float (*red)[4], int tid) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[wv][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < MM_NW; ++w) t += red[w][k];                // fixed order
        v[k] = t;
    }
}

template <int NT, int NCOL, bool FIN>
__global__ __launch_bounds__(MM_THREADS) void chan_matmul_fwd16_kernel(const fgnn_slab16 ya, const fgnn_slab16 yb,
                                                                       const int *nvalid, int N, int ldr, int G, void *out,
                                                                       long long ogstride, long long ldo, const FinArgs16 F) {
    extern __shared__ __attribute__((aligned(16))) char mm_lds[];
    __shared__ float fin_red[MM_NW][4];
    constexpr bool STRIP = MMCfg<NT, NCOL>::STRIP;
    const int C = ya.C, gc = blockIdx.x, g = gc / C, c = gc - g * C, tid = threadIdx.x;
    young_prio(5, tid >> 6, MM_NW);
    const int nv = nvalid_of(nvalid, g, N), ntv = (nv + 31) / 32;
    Src16 A, B;
    u32x4 pre_a[4], pre_b[4];
    if constexpr (FIN) {
        A = mm_src_plain(ya.ptr, ya.gstride, ya.ldp, G, g, c);
        B = mm_src_plain(yb.ptr, yb.gstride, yb.ldp, G, g, c);
        A.norm = B.norm = true;
        // the tile statistics first (small, needed first), then the first operand chunk behind them
        constexpr int TPT = 1024 / MM_THREADS;                           // tpg <= 1024 (N <= 256)
        float n[TPT];
        float2 pa[TPT], pb[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int t = tid + k * MM_THREADS, tc = t < F.tpg ? t : 0;
            n[k] = t < F.tpg ? F.cnt[(long long)g * F.tpg + tc] : 0.f;
            pa[k] = reinterpret_cast<const float2 *>(F.part_a)[((long long)g * C + c) * F.tpg + tc];
            pb[k] = reinterpret_cast<const float2 *>(F.part_b)[((long long)g * C + c) * F.tpg + tc];
            if (t >= F.tpg) pa[k].y = pb[k].y = 0.f;
        }
        if constexpr (STRIP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pre_a[i] = StageXK<NT>::load_piece(i, A, ldr, nv, 0, tid);
                pre_b[i] = StageKX<NT>::load_piece(i, B, ldr, nv, 0, tid);
            }
        }
        float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            s[0] += n[k];
            s[1] += n[k] * pa[k].x;
            s[2] += n[k] * pb[k].x;
        }
        wg_sum<3>(s, fin_red, tid);
        const float ma = s[0] > 0.f ? s[1] / s[0] : 0.f, mb = s[0] > 0.f ? s[2] / s[0] : 0.f;
        float q[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const float da = pa[k].x - ma, db = pb[k].x - mb;
            q[0] += pa[k].y + n[k] * da * da;
            q[1] += pb[k].y + n[k] * db * db;
        }
        wg_sum<2>(q, fin_red, tid);
        const float4 ra = nrm_record(ma, q[0], s[0], (float)nv, F.gw_a ? F.gw_a[c] : 1.f, F.eps);
        const float4 rb = nrm_record(mb, q[1], s[0], (float)nv, F.gw_b ? F.gw_b[c] : 1.f, F.eps);
        if (tid == 0) {
            reinterpret_cast<float4 *>(F.nrm_a)[gc] = ra;
            reinterpret_cast<float4 *>(F.nrm_b)[gc] = rb;
        }
        auto sgpr = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
        A.a = sgpr(ra.y);
        A.mean = sgpr(ra.x);
        A.b = sgpr((ya.beta ? ya.beta[c] : 0.f) - ra.x * ra.y);
        B.a = sgpr(rb.y);
        B.mean = sgpr(rb.x);
        B.b = sgpr((yb.beta ? yb.beta[c] : 0.f) - rb.x * rb.y);
    } else {
        A = mm_src(ya, G, g, c);
        B = mm_src(yb, G, g, c);
    }
    const View16 vO = make_view16(out, ogstride, ldo, G);
    const int o_off = g * vO.gs2 + c * vO.ld2;
    f32x16 acc[MMCfg<NT, NCOL>::MAXT];
    mm_gemm<NT, NCOL, true, false, false, false, FIN && STRIP>(acc, A, B, mm_lds, ldr, nv, ntv, tid, pre_a, pre_b);   // M = Ya Yb
    float s1 = 0.f, s2 = 0.f;
    mm_store<NT, NCOL, 0>(acc, mm_lds, vO, o_off, A, N, ldr, nv, ntv, s1, s2, tid);
}

// STATS: 0 none; 1 S1/S2 of both outputs from the raw operand slabs (re-read in the store epilogue); 2 from the trace term:
//   sum dA (z_a - mean_a) = (T - beta_a S1_a) / a_a   and   sum dB (z_b - mean_b) = (T - beta_b S1_b) / a_b,
//   T = <dM, Ya Yb> = sum over the tiles of tpart[g][c][.]  (emitted by the kernel that produced dM: fgnn_mlp_bwd16 of mlp3, whose
//   first slab is the forward product M).  Both identities are exact for the un-rounded products; with the stored bf16 values
//   they carry the same 2^-9 / sqrt(3) relative noise as the sums of mode 1, and the two raw slabs are not read again.
template <int NT, int NCOL, int STATS>
__global__ __launch_bounds__(MM_THREADS) void chan_matmul_bwd16_kernel(const fgnn_slab16 ya, const fgnn_slab16 yb,
                                                                       const void *dm, long long dmg, long long ldm,
                                                                       const int *nvalid, int N, int ldr, int G, void *da,
                                                                       void *db, long long ogstride, long long ldo,
                                                                       float *s12a, float *s12b, const float *tpart, int tpg,
                                                                       float *coefa, float *coefb) {
    extern __shared__ __attribute__((aligned(16))) char mm_lds[];
    __shared__ float red[MM_NW][4];
    __shared__ float t_keep;
    constexpr bool STRIP = MMCfg<NT, NCOL>::STRIP;
    const int C = ya.C, gc = blockIdx.x, g = gc / C, c = gc - g * C, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int nv = nvalid_of(nvalid, g, N), ntv = (nv + 31) / 32;
    young_prio(4, wv, MM_NW);
    const Src16 A = mm_src(ya, G, g, c), B = mm_src(yb, G, g, c);
    const Src16 D = mm_src_plain(dm, dmg, ldm, G, g, c);
    const View16 vOA = make_view16(da, ogstride, ldo, G), vOB = make_view16(db, ogstride, ldo, G);
    const int o_off = g * vOA.gs2 + c * vOA.ld2;
    float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
    u32x4 pre_a[4], pre_b[4];
    constexpr bool PRE = STATS == 2 && STRIP;
    if constexpr (STATS == 2) {
        // tile partials first, then the first operand chunk behind them, then the (barrier-bound) reduction
        constexpr int TPT = 1024 / MM_THREADS;
        float tp[1] = {0.f};
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int t = tid + k * MM_THREADS;
            const float p = tpart[((long long)g * C + c) * tpg + (t < tpg ? t : 0)];
            tp[0] += t < tpg ? p : 0.f;
        }
        if constexpr (STRIP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pre_a[i] = StageXK<NT>::load_piece(i, D, ldr, nv, 0, tid);
                pre_b[i] = StageXK<NT>::load_piece(i, B, ldr, nv, 0, tid);
            }
        }
        wg_sum<1>(tp, red, tid);
        if (tid == 0) t_keep = tp[0];    // parked in LDS: nothing below may hold a register across the two products
        __syncthreads();                 // `red` is used again at the end
    }
    {
        f32x16 acc[MMCfg<NT, NCOL>::MAXT];
        // the first product's last chunk already requests chunk 0 of the second one (both of its operands are [k][x] sources)
        auto next = [&](int i, u32x4 &ra, u32x4 &rb) {
            if constexpr (STRIP) {
                ra = StageKX<NT>::load_piece(i, A, ldr, nv, 0, tid);
                rb = StageKX<NT>::load_piece(i, D, ldr, nv, 0, tid);
            }
        };
        mm_gemm<NT, NCOL, true, true, true, false, PRE>(acc, D, B, mm_lds, ldr, nv, ntv, tid, pre_a, pre_b, next);     // dA = dM Yb^T
        mm_store<NT, NCOL, STATS>(acc, mm_lds, vOA, o_off, A, N, ldr, nv, ntv, sa1, sa2, tid);
        if constexpr (STATS != 0) {       // the sums of dA leave the registers before the second product starts
            sa1 = wave_sum(sa1);
            sa2 = wave_sum(sa2);
            if (lane == 0) {
                red[wv][0] = sa1;
                red[wv][1] = sa2;
            }
        }
        mm_gemm<NT, NCOL, false, false, false, true, true>(acc, A, D, mm_lds, ldr, nv, ntv, tid, pre_a, pre_b);         // dB = Ya^T dM
        mm_store<NT, NCOL, STATS>(acc, mm_lds, vOB, o_off, B, N, ldr, nv, ntv, sb1, sb2, tid);
    }
    if constexpr (STATS != 0) {
        sb1 = wave_sum(sb1);
        sb2 = wave_sum(sb2);
        if (lane == 0) {
            red[wv][2] = sb1;
            red[wv][3] = sb2;
        }
        __syncthreads();
        if (tid < 4) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < MM_NW; ++w) v += red[w][tid];                      // fixed order
            if constexpr (STATS == 2) {
                // lanes 0 / 2 hold S1 of dA / dB; lanes 1 / 3 form S2 from T and the S1 of their neighbour
                const float T = t_keep;
                const float s1 = __shfl(v, tid & ~1);
                const Src16 &O = tid < 2 ? A : B;
                const float beta = O.b + O.mean * O.a;                              // b = beta - mean a
                if (tid & 1) v = O.a != 0.f ? (T - beta * s1) / O.a : 0.f;
            }
            float *dst = (tid < 2 ? s12a : s12b) + (long long)gc * 2 + (tid & 1);
            *dst = v;
            // dz coefficients of the two producing MLPs (the arithmetic of gn_bwd_coef_kernel), saving that launch
            const float v2 = __shfl(v, tid | 1);
            float *cf = tid < 2 ? coefa : coefb;
            if (cf && !(tid & 1)) {
                const float4 n = reinterpret_cast<const float4 *>(tid < 2 ? ya.nrm : yb.nrm)[gc];
                const float m = (float)nv * (float)nv;
                float4 o;
                o.x = n.x;
                o.y = n.y;
                o.z = m > 0.f ? -n.y * v2 * n.w / m : 0.f;
                o.w = m > 0.f ? -n.y * v / m : 0.f;
                reinterpret_cast<float4 *>(cf)[gc] = o;
            }
        }
    }
}

template <int NT, int NCOL, bool FIN>
int launch_fwd16_impl(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr, void *out,
                      long long ogstride, long long ldo, const FinArgs16 &F, hipStream_t st) {
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)chan_matmul_fwd16_kernel<NT, NCOL, FIN>, MMCfg<NT>::LDS_BYTES);
    hipLaunchKernelGGL((chan_matmul_fwd16_kernel<NT, NCOL, FIN>), dim3(G * ya->C), dim3(MM_THREADS), MMCfg<NT>::LDS_BYTES, st, *ya,
                       *yb, nvalid, N, ldr, G, out, ogstride, ldo, F);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int NT, int NCOL = NT>
int launch_fwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr, void *out,
                 long long ogstride, long long ldo, const FinArgs16 *F, hipStream_t st) {
    if (F) return launch_fwd16_impl<NT, NCOL, true>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, *F, st);
    return launch_fwd16_impl<NT, NCOL, false>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, FinArgs16{}, st);
}
struct CoefOut {
    float *a = nullptr, *b = nullptr;       // optional (G*C*4) dz-coefficient records of the two operand MLPs
};
template <int NT, int NCOL, int STATS>
int launch_bwd16_impl(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmg, long long ldm, const int *nvalid,
                      int G, int N, int ldr, void *da, void *db, long long ogstride, long long ldo, float *s12a, float *s12b,
                      const float *tpart, int tpg, hipStream_t st, CoefOut co) {
    static LdsAttrCache attr_cache;
    (void)fgnn_raise_lds(attr_cache, (const void *)chan_matmul_bwd16_kernel<NT, NCOL, STATS>, MMCfg<NT>::LDS_BYTES);
    hipLaunchKernelGGL((chan_matmul_bwd16_kernel<NT, NCOL, STATS>), dim3(G * ya->C), dim3(MM_THREADS), MMCfg<NT>::LDS_BYTES, st, *ya,
                       *yb, dm, dmg, ldm, nvalid, N, ldr, G, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, co.a, co.b);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int NT, int NCOL = NT>
int launch_bwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmg, long long ldm, const int *nvalid,
                 int G, int N, int ldr, void *da, void *db, long long ogstride, long long ldo, float *s12a, float *s12b,
                 const float *tpart, int tpg, hipStream_t st, CoefOut co) {
    if (!s12a) return launch_bwd16_impl<NT, NCOL, 0>(ya, yb, dm, dmg, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
    if (tpart) return launch_bwd16_impl<NT, NCOL, 2>(ya, yb, dm, dmg, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
    return launch_bwd16_impl<NT, NCOL, 1>(ya, yb, dm, dmg, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
}

}  // namespace

#if MM_ABLATE == 5
extern "C" int fgnn_debug_mm16_stamps(unsigned long long *host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(mm_stamps), sizeof(mm_stamps));
}
#endif

static int matmul_fwd16_common(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr, void *out,
                               long long ogstride, long long ldo, const FinArgs16 *F, void *stream) {
    FGNN_CHECK(ya && yb && out && ya->ptr && yb->ptr, "fgnn_chan_matmul_fwd16: null argument");
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0 && N <= 256, "fgnn_chan_matmul_fwd16: bad shapes (N <= 256)");
    FGNN_CHECK(ldr >= N && ldr % 8 == 0, "fgnn_chan_matmul_fwd16: ldr must be a multiple of 8 and >= N");
    FGNN_CHECK(ya->ldp % 8 == 0 && yb->ldp % 8 == 0 && ya->gstride % 8 == 0 && yb->gstride % 8 == 0,
               "fgnn_chan_matmul_fwd16: channel / graph strides must be multiples of 8 elements (16-byte loads)");
    FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 2 && (long long)G * yb->gstride < 0x7fffffffll / 2 &&
               (long long)G * ogstride < 0x7fffffffll / 2,
               "fgnn_chan_matmul_fwd16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return launch_fwd16<2>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, F, st);
    if (N <= 128) return launch_fwd16<4>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, F, st);
    if (N <= 224) return launch_fwd16<8, 7>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, F, st);   // 7 tile columns: 16 registers less
    return launch_fwd16<8>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, F, st);
}

extern "C" int fgnn_chan_matmul_fwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr,
                                      void *out, long long ogstride, long long ldo, void *stream) {
    return matmul_fwd16_common(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, nullptr, stream);
}

extern "C" int fgnn_chan_matmul_fwd16_fin(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const float *part_a, const float *part_b,
                                          const float *cnt, const float *gn_weight_a, const float *gn_weight_b, float eps,
                                          int tpg, const int *nvalid, int G, int N, int ldr, void *out, long long ogstride,
                                          long long ldo, void *stream) {
    FGNN_CHECK(ya && yb && part_a && part_b && cnt && ya->nrm && yb->nrm && tpg > 0 && tpg <= 1024,
               "fgnn_chan_matmul_fwd16_fin: needs the tile statistics, the record buffers (slab.nrm) and tpg <= 1024");
    FinArgs16 F;
    F.part_a = part_a;
    F.part_b = part_b;
    F.cnt = cnt;
    F.gw_a = gn_weight_a;
    F.gw_b = gn_weight_b;
    F.nrm_a = const_cast<float *>(ya->nrm);
    F.nrm_b = const_cast<float *>(yb->nrm);
    F.eps = eps;
    F.tpg = tpg;
    return matmul_fwd16_common(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, &F, stream);
}

static int matmul_bwd16_common(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride, long long ldm,
                               const int *nvalid, int G, int N, int ldr, void *da, void *db, long long ogstride, long long ldo,
                               float *s12a, float *s12b, const float *tpart, int tpg, void *stream, float *coefa = nullptr,
                               float *coefb = nullptr) {
    FGNN_CHECK(ya && yb && dm && da && db && ya->ptr && yb->ptr, "fgnn_chan_matmul_bwd16: null argument");
    FGNN_CHECK((!coefa && !coefb) || (s12a && coefa && coefb), "fgnn_chan_matmul_bwd16_tc: the coefficient records come with the s12 sums");
    CoefOut co;
    co.a = coefa;
    co.b = coefb;
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0 && N <= 256, "fgnn_chan_matmul_bwd16: bad shapes (N <= 256)");
    FGNN_CHECK(ldr >= N && ldr % 8 == 0, "fgnn_chan_matmul_bwd16: ldr must be a multiple of 8 and >= N");
    FGNN_CHECK(ya->ldp % 8 == 0 && yb->ldp % 8 == 0 && ya->gstride % 8 == 0 && yb->gstride % 8 == 0 && ldm % 8 == 0 &&
               dmgstride % 8 == 0,
               "fgnn_chan_matmul_bwd16: channel / graph strides must be multiples of 8 elements (16-byte loads)");
    FGNN_CHECK((s12a == nullptr) == (s12b == nullptr), "fgnn_chan_matmul_bwd16: s12a and s12b come together");
    FGNN_CHECK(!s12a || (ya->nrm && yb->nrm), "fgnn_chan_matmul_bwd16: s12 outputs need normalised slabs");
    FGNN_CHECK(!tpart || (s12a && tpg > 0 && tpg <= 1024), "fgnn_chan_matmul_bwd16_t: tile partials need the s12 outputs and tpg <= 1024");
    FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 2 && (long long)G * yb->gstride < 0x7fffffffll / 2 &&
               (long long)G * dmgstride < 0x7fffffffll / 2 && (long long)G * ogstride < 0x7fffffffll / 2,
               "fgnn_chan_matmul_bwd16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return launch_bwd16<2>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
    if (N <= 128) return launch_bwd16<4>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
    if (N <= 224)
        return launch_bwd16<8, 7>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
    return launch_bwd16<8>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, st, co);
}

extern "C" int fgnn_chan_matmul_bwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride,
                                      long long ldm, const int *nvalid, int G, int N, int ldr, void *da, void *db,
                                      long long ogstride, long long ldo, float *s12a, float *s12b, void *stream) {
    return matmul_bwd16_common(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, nullptr, 0, stream);
}

extern "C" int fgnn_chan_matmul_bwd16_t(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride,
                                        long long ldm, const float *tpart, int tpg, const int *nvalid, int G, int N, int ldr,
                                        void *da, void *db, long long ogstride, long long ldo, float *s12a, float *s12b,
                                        void *stream) {
    FGNN_CHECK(tpart != nullptr, "fgnn_chan_matmul_bwd16_t: missing tile partials");
    return matmul_bwd16_common(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, stream);
}

extern "C" int fgnn_chan_matmul_bwd16_tc(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride,
                                         long long ldm, const float *tpart, int tpg, const int *nvalid, int G, int N, int ldr,
                                         void *da, void *db, long long ogstride, long long ldo, float *s12a, float *s12b,
                                         float *coefa, float *coefb, void *stream) {
    FGNN_CHECK(tpart != nullptr, "fgnn_chan_matmul_bwd16_tc: missing tile partials");
    return matmul_bwd16_common(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, tpart, tpg, stream,
                               coefa, coefb);
}
