// Per-channel N x N matrix products on bf16 slabs (Matmul, models/layers.py:161-162) and their backward for gfx950,
// N <= 256.  One 512-thread workgroup owns one (g,c) matrix: every operand element is read from HBM once, normalised on
// load ((z - mean) a + beta in fp32, padding -> 0), rounded to bf16 and streamed through double-buffered LDS panels in
// 64-wide k chunks; all ceil(nv/32)^2 output tiles of 32x32 live in the fp32 accumulators of the eight waves and are
// multiplied with v_mfma_f32_32x32x16_bf16.
//   Out[m][n] = sum_k OpA(m,k) OpB(k,n),  OpA(m,k) = A_KC ? MA[m][k] : MA[k][m],  OpB(k,n) = B_KC ? MB[n][k] : MB[k][n]
//   forward M = Ya Yb: (KC, KR);   dA = dM Yb^T: (KC, KC);   dB = Ya^T dM: (KR, KR)
// Every panel is stored [x][kk] (k contiguous, 68-element = 136-byte row stride: an MFMA operand of one k-step is two
// conflict-free ds_read_b64).  A K-contiguous source is copied row by row (8-byte loads); a K-row source is transposed
// on the way in: a lane loads the pixel pair (x, x+1) of four consecutive source rows and writes two 8-byte panel rows.
// Matrix rows are `ldr` elements apart (ldr % 8 == 0), so all of these accesses are aligned.
#include "fgnn_bf16.h"

#ifndef MM_ABLATE
#define MM_ABLATE 0          // debug builds (tools/gpu_mm16_ablate.py): 1 no stores, 2 no MFMA, 3 no LDS staging, 4 no global loads
#endif

namespace {

constexpr int MM_NW = 8, MM_THREADS = 64 * MM_NW, MM_KC = 64, MM_LDK = 68;   // panel row stride in elements

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
    // tile ownership: NT == 8 (N > 128): wave w owns the tile ROW w (all tile columns): its A-operand fragment of a k-step is
    // read from LDS once and reused for every column (2.3 instead of 4 ds_read_b64 per MFMA) and no index arithmetic is
    // left in the k loop; smaller matrices (few tile rows) keep the cyclic assignment tile = wave + 8 i
    static constexpr bool STRIP = NT == 8;
    static constexpr int MAXT = STRIP ? NCOL : (NT * NT + MM_NW - 1) / MM_NW;
    static constexpr int PANEL_B = XM * MM_LDK * 2;                 // bytes of one panel
    static constexpr int BUF_B = 2 * PANEL_B;                       // A + B panel
    static constexpr int LDS_BYTES = 2 * BUF_B;                     // double buffered
    static constexpr int KC_SWEEPS = XM / 32;                       // 8-byte pieces per thread per K-contiguous panel
    static constexpr int KR_SWEEPS = (16 * (XM / 2)) / MM_THREADS;  // 4x2 micro-tiles per thread per K-row panel
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
        o.a = n.y;
        o.b = be - n.x * n.y;
        o.mean = n.x;
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

// ---- K-contiguous source: thread -> (row x = tid/16 + 32*sweep, 4 k's at k0 + 4*(tid%16)) ------------------------------
template <int NT>
DEVI void kc_load(uint2 (&x)[MMCfg<NT>::KC_SWEEPS], const Src16 &s, int ldr, int nv, int k0, int tid) {
    const int kq = tid & 15, xr = tid >> 4;
    const int kk = k0 + 4 * kq;
    const int base = kk < nv ? (xr * ldr + kk) * 2 : OOB_OFF;
#pragma unroll
    for (int i = 0; i < MMCfg<NT>::KC_SWEEPS; ++i) {
        const int off = (xr + 32 * i) < nv ? base : OOB_OFF;
        const int so = s.off2 + 32 * i * ldr * 2;
        x[i].x = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(s.v.r, off, so, 0);
        x[i].y = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(s.v.r, off, so + 4, 0);
    }
}
template <int NT>
DEVI void kc_stage(char *P, const uint2 (&x)[MMCfg<NT>::KC_SWEEPS], const Src16 &s, int nv, int k0, int tid) {
    const int kq = tid & 15, xr = tid >> 4;
    const int kk = k0 + 4 * kq;
    char *dst = P + xr * (MM_LDK * 2) + kq * 8;
    if (s.norm && (nv & 3) == 0) {
        // nv % 4 == 0: a thread's four k's are valid or padding together -> the mask folds into the affine pair
        const bool kok = kk < nv;
#pragma unroll
        for (int i = 0; i < MMCfg<NT>::KC_SWEEPS; ++i) {
            const bool ok = kok && (xr + 32 * i) < nv;
            const float a = ok ? s.a : 0.f, b = ok ? s.b : 0.f;
            uint2 o;
            o.x = cvt_pk(fmaf(bf_lo(x[i].x), a, b), fmaf(bf_hi(x[i].x), a, b));
            o.y = cvt_pk(fmaf(bf_lo(x[i].y), a, b), fmaf(bf_hi(x[i].y), a, b));
            *reinterpret_cast<uint2 *>(dst + 32 * i * (MM_LDK * 2)) = o;
        }
        return;
    }
    float m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = (kk + e) < nv ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < MMCfg<NT>::KC_SWEEPS; ++i) {
        uint2 o = x[i];
        if (s.norm) {
            const float r = (xr + 32 * i) < nv ? 1.f : 0.f;
            o.x = cvt_pk(fmaf(bf_lo(x[i].x), s.a, s.b) * (m[0] * r), fmaf(bf_hi(x[i].x), s.a, s.b) * (m[1] * r));
            o.y = cvt_pk(fmaf(bf_lo(x[i].y), s.a, s.b) * (m[2] * r), fmaf(bf_hi(x[i].y), s.a, s.b) * (m[3] * r));
        }
        *reinterpret_cast<uint2 *>(dst + 32 * i * (MM_LDK * 2)) = o;
    }
}

// ---- K-row source: thread -> (pixel pair xp = tid % (XM/2), k group kg = tid / (XM/2) + (MM_THREADS/(XM/2)) * sweep) -----
template <int NT>
DEVI void kr_load(unsigned (&x)[MMCfg<NT>::KR_SWEEPS][4], const Src16 &s, int ldr, int nv, int k0, int tid) {
    constexpr int HX = MMCfg<NT>::XM / 2, KGS = MM_THREADS / HX;
    const int xp = tid % HX, kg0 = tid / HX;
    const int base = 2 * xp < nv ? 4 * xp : OOB_OFF;
#pragma unroll
    for (int i = 0; i < MMCfg<NT>::KR_SWEEPS; ++i) {
        const int kr = k0 + 4 * (kg0 + KGS * i);                     // wave-uniform when HX >= 64
#pragma unroll
        for (int r = 0; r < 4; ++r)
            x[i][r] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(s.v.r, (kr + r) < nv ? base : OOB_OFF,
                                                                      s.off2 + (kr + r) * ldr * 2, 0);
    }
}
template <int NT>
DEVI void kr_stage(char *P, const unsigned (&x)[MMCfg<NT>::KR_SWEEPS][4], const Src16 &s, int nv, int k0, int tid) {
    constexpr int HX = MMCfg<NT>::XM / 2, KGS = MM_THREADS / HX;
    const int xp = tid % HX, kg0 = tid / HX;
    const float c0 = 2 * xp < nv ? 1.f : 0.f, c1 = (2 * xp + 1) < nv ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < MMCfg<NT>::KR_SWEEPS; ++i) {
        const int kg = kg0 + KGS * i;
        const int kr = k0 + 4 * kg;
        uint2 lo, hi;
        if (s.norm && (nv & 3) == 0) {
            // nv % 4 == 0: the 4 x 2 micro-tile is valid or padding as a whole
            const bool ok = kr < nv && 2 * xp < nv;
            const float a = ok ? s.a : 0.f, b = ok ? s.b : 0.f;
            lo.x = cvt_pk(fmaf(bf_lo(x[i][0]), a, b), fmaf(bf_lo(x[i][1]), a, b));
            lo.y = cvt_pk(fmaf(bf_lo(x[i][2]), a, b), fmaf(bf_lo(x[i][3]), a, b));
            hi.x = cvt_pk(fmaf(bf_hi(x[i][0]), a, b), fmaf(bf_hi(x[i][1]), a, b));
            hi.y = cvt_pk(fmaf(bf_hi(x[i][2]), a, b), fmaf(bf_hi(x[i][3]), a, b));
        } else if (s.norm) {
            float f0[4], f1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m = (kr + r) < nv ? 1.f : 0.f;
                f0[r] = fmaf(bf_lo(x[i][r]), s.a, s.b) * (m * c0);
                f1[r] = fmaf(bf_hi(x[i][r]), s.a, s.b) * (m * c1);
            }
            lo.x = cvt_pk(f0[0], f0[1]);
            lo.y = cvt_pk(f0[2], f0[3]);
            hi.x = cvt_pk(f1[0], f1[1]);
            hi.y = cvt_pk(f1[2], f1[3]);
        } else {
            lo.x = pack_lo(x[i][0], x[i][1]);
            lo.y = pack_lo(x[i][2], x[i][3]);
            hi.x = pack_hi(x[i][0], x[i][1]);
            hi.y = pack_hi(x[i][2], x[i][3]);
        }
        char *dst = P + (2 * xp) * (MM_LDK * 2) + kg * 8;
        *reinterpret_cast<uint2 *>(dst) = lo;
        *reinterpret_cast<uint2 *>(dst + MM_LDK * 2) = hi;
    }
}

template <int NT, bool KC>
struct Stage {
    uint2 c[KC ? MMCfg<NT>::KC_SWEEPS : 1];
    unsigned r[KC ? 1 : MMCfg<NT>::KR_SWEEPS][4];
    DEVI void load(const Src16 &s, int ldr, int nv, int k0, int tid) {
        if (MM_ABLATE == 4) {
            for (auto &v : c) v = make_uint2(tid, k0);
            for (auto &v : r) v[0] = v[1] = v[2] = v[3] = tid;
            return;
        }
        if constexpr (KC) kc_load<NT>(c, s, ldr, nv, k0, tid);
        else kr_load<NT>(r, s, ldr, nv, k0, tid);
    }
    DEVI void stage(char *P, const Src16 &s, int nv, int k0, int tid) {
        if (MM_ABLATE == 3) {
            for (auto &v : c) asm volatile("" ::"v"(v.x), "v"(v.y));
            for (auto &v : r) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
            return;
        }
        if constexpr (KC) kc_stage<NT>(P, c, s, nv, k0, tid);
        else kr_stage<NT>(P, r, s, nv, k0, tid);
    }
};

// one k-step (16 k's) of the 32-row strip `t` of a panel for lane (j, h): k = 16*step + 8h .. +7
DEVI i32x4 panel_operand(const char *P, int t, int step, int j, int h) {
    const uint2 *p = reinterpret_cast<const uint2 *>(P + (32 * t + j) * (MM_LDK * 2) + (16 * step + 8 * h) * 2);
    const uint2 a = p[0], b = p[1];
    i32x4 o;
    o[0] = (int)a.x;
    o[1] = (int)a.y;
    o[2] = (int)b.x;
    o[3] = (int)b.y;
    return o;
}

// acc[ti] (tile wv + 8*ti of the ntv x ntv valid tiles) += OpA OpB over all k chunks
template <int NT, int NCOL, bool A_KC, bool B_KC>
DEVI void mm_gemm(AccArray<NT, NCOL> &acc, const Src16 &A, const Src16 &B, char *lds, int ldr, int nv, int ntv,
                  int tid) {
    using Cf = MMCfg<NT, NCOL>;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int T = ntv * ntv;
#pragma unroll
    for (int ti = 0; ti < Cf::MAXT; ++ti) zero16f(acc[ti]);
    const int nkc = (nv + MM_KC - 1) / MM_KC;
    // Two register sets per operand: the loads of chunk c + 2 are issued as soon as the set that held chunk c has been staged,
    // so two chunks (2 x 50 KB per CU at N = 200) are in flight while one is multiplied -- with a single set every CU of the
    // chip alternated in lockstep between "everyone waits for its chunk" and "everyone multiplies with HBM idle".
    Stage<NT, A_KC> sa[2];
    Stage<NT, B_KC> sb[2];
    auto compute = [&](const char *pa) {
        if (MM_ABLATE == 2) return;
        const char *pb = pa + Cf::PANEL_B;
        if constexpr (Cf::STRIP) {
            if (wv < ntv) {
#pragma unroll
                for (int s = 0; s < MM_KC / 16; ++s) {
                    const i32x4 a = panel_operand(pa, wv, s, j, h);
#pragma unroll
                    for (int tn = 0; tn < NCOL; ++tn)
                        if (tn < ntv) acc[tn] = mfma16(a, panel_operand(pb, tn, s, j, h), acc[tn]);
                }
            }
        } else {
#pragma unroll
            for (int ti = 0; ti < Cf::MAXT; ++ti) {
                const int t = wv + MM_NW * ti;
                if (t < T) {
                    const int tm = t / ntv, tn = t - tm * ntv;
#pragma unroll
                    for (int s = 0; s < MM_KC / 16; ++s)
                        acc[ti] = mfma16(panel_operand(pa, tm, s, j, h), panel_operand(pb, tn, s, j, h), acc[ti]);
                }
            }
        }
    };
    sa[0].load(A, ldr, nv, 0, tid);
    sb[0].load(B, ldr, nv, 0, tid);
    if (nkc > 1) {
        sa[1].load(A, ldr, nv, MM_KC, tid);
        sb[1].load(B, ldr, nv, MM_KC, tid);
    }
    sa[0].stage(lds, A, nv, 0, tid);
    sb[0].stage(lds + Cf::PANEL_B, B, nv, 0, tid);
    if (nkc > 2) {
        sa[0].load(A, ldr, nv, 2 * MM_KC, tid);
        sb[0].load(B, ldr, nv, 2 * MM_KC, tid);
    }
    __syncthreads();
    for (int c = 0; c < nkc; c += 2) {
        // even chunk c in buffer 0; chunk c + 1 (register set 1) goes to buffer 1 first, then its set is re-loaded with c + 3
        if (c + 1 < nkc) {
            sa[1].stage(lds + Cf::BUF_B, A, nv, (c + 1) * MM_KC, tid);
            sb[1].stage(lds + Cf::BUF_B + Cf::PANEL_B, B, nv, (c + 1) * MM_KC, tid);
            if (c + 3 < nkc) {
                sa[1].load(A, ldr, nv, (c + 3) * MM_KC, tid);
                sb[1].load(B, ldr, nv, (c + 3) * MM_KC, tid);
            }
        }
        compute(lds);
        __syncthreads();
        if (c + 1 < nkc) {
            if (c + 2 < nkc) {
                sa[0].stage(lds, A, nv, (c + 2) * MM_KC, tid);
                sb[0].stage(lds + Cf::PANEL_B, B, nv, (c + 2) * MM_KC, tid);
                if (c + 4 < nkc) {
                    sa[0].load(A, ldr, nv, (c + 4) * MM_KC, tid);
                    sb[0].load(B, ldr, nv, (c + 4) * MM_KC, tid);
                }
            }
            compute(lds + Cf::BUF_B);
            __syncthreads();
        }
    }
}

// accumulators -> global as bf16 (rows >= N / columns >= ldr dropped), optionally S1 = sum t, S2 = sum t * (raw - mean)
// of the ROUNDED values over the valid entries, `raw` re-read from the un-normalised slab
template <int NT, int NCOL, bool WANT_S>
DEVI void mm_store(const AccArray<NT, NCOL> &acc, const View16 &ov, int o_off2, const Src16 &raw, int N, int ldr,
                   int nv, int ntv, float &s1, float &s2, int tid) {
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int T = ntv * ntv;
#pragma unroll
    for (int ti = 0; ti < MMCfg<NT, NCOL>::MAXT; ++ti) {
        const int t = MMCfg<NT, NCOL>::STRIP ? (wv < ntv && ti < ntv ? wv * ntv + ti : T) : wv + MM_NW * ti;
        if (t < T) {
            const int tm = MMCfg<NT, NCOL>::STRIP ? wv : t / ntv, tn = MMCfg<NT, NCOL>::STRIP ? ti : t - tm * ntv;
            const int col = 32 * tn + j;
            const int rowb = 32 * tm + 4 * h;
            const int base = col < ldr ? (rowb * ldr + col) * 2 : OOB_OFF;
            const int vbase = col < nv ? (rowb * ldr + col) * 2 : OOB_OFF;
            unsigned u[16];
            if (WANT_S) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    u[r] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(raw.v.r, (rowb + dr) < nv ? vbase : OOB_OFF,
                                                                           raw.off2 + dr * ldr * 2, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                const unsigned d = cvt_pk(acc[ti][r], 0.f);
                if (MM_ABLATE == 1) {
                    asm volatile("" ::"v"(d));
                    continue;
                }
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(d & 0xffffu), ov.r, (rowb + dr) < N ? base : OOB_OFF,
                                                      o_off2 + dr * ldr * 2, 0);
                if (WANT_S) {
                    const float m = ((rowb + dr) < nv && col < nv) ? 1.f : 0.f;
                    const float tv = bf_lo(d) * m;
                    s1 += tv;
                    s2 += tv * (bf_lo(u[r]) - raw.mean);
                }
            }
        }
    }
}

// zero the part of an N x ldr output outside the first X = 32*ntv rows / columns (pixel pairs; X is even)
DEVI void mm_zero_fill(const View16 &ov, int o_off2, int N, int ldr, int X, int tid) {
    if (X >= N && X >= ldr) return;
    const int hp = ldr / 2;
    for (int q = tid; q < N * hp; q += MM_THREADS) {
        const int r = q / hp, cp = q - r * hp;
        if (r >= X || 2 * cp >= X) buf_store_u32(0u, ov, q * 4, o_off2);
    }
}

template <int NT, int NCOL>
__global__ __launch_bounds__(MM_THREADS) void chan_matmul_fwd16_kernel(const fgnn_slab16 ya, const fgnn_slab16 yb,
                                                                       const int *nvalid, int N, int ldr, int G, void *out,
                                                                       long long ogstride, long long ldo) {
    extern __shared__ __attribute__((aligned(16))) char mm_lds[];
    const int C = ya.C, gc = blockIdx.x, g = gc / C, c = gc - g * C, tid = threadIdx.x;
    const int nv = nvalid_of(nvalid, g, N), ntv = (nv + 31) / 32;
    const Src16 A = mm_src(ya, G, g, c), B = mm_src(yb, G, g, c);
    const View16 vO = make_view16(out, ogstride, ldo, G);
    const int o_off = g * vO.gs2 + c * vO.ld2;
    mm_zero_fill(vO, o_off, N, ldr, 32 * ntv, tid);
    if (ntv == 0) return;
    f32x16 acc[MMCfg<NT, NCOL>::MAXT];
    mm_gemm<NT, NCOL, true, false>(acc, A, B, mm_lds, ldr, nv, ntv, tid);
    float s1 = 0.f, s2 = 0.f;
    mm_store<NT, NCOL, false>(acc, vO, o_off, A, N, ldr, nv, ntv, s1, s2, tid);
}

template <int NT, int NCOL>
__global__ __launch_bounds__(MM_THREADS) void chan_matmul_bwd16_kernel(const fgnn_slab16 ya, const fgnn_slab16 yb,
                                                                       const void *dm, long long dmg, long long ldm,
                                                                       const int *nvalid, int N, int ldr, int G, void *da,
                                                                       void *db, long long ogstride, long long ldo,
                                                                       float *s12a, float *s12b) {
    extern __shared__ __attribute__((aligned(16))) char mm_lds[];
    __shared__ float red[MM_NW][4];
    const int C = ya.C, gc = blockIdx.x, g = gc / C, c = gc - g * C, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int nv = nvalid_of(nvalid, g, N), ntv = (nv + 31) / 32;
    const Src16 A = mm_src(ya, G, g, c), B = mm_src(yb, G, g, c);
    const Src16 D = mm_src_plain(dm, dmg, ldm, G, g, c);
    const View16 vOA = make_view16(da, ogstride, ldo, G), vOB = make_view16(db, ogstride, ldo, G);
    const int o_off = g * vOA.gs2 + c * vOA.ld2;
    mm_zero_fill(vOA, o_off, N, ldr, 32 * ntv, tid);
    mm_zero_fill(vOB, o_off, N, ldr, 32 * ntv, tid);
    float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
    if (ntv > 0) {
        f32x16 acc[MMCfg<NT, NCOL>::MAXT];
        mm_gemm<NT, NCOL, true, true>(acc, D, B, mm_lds, ldr, nv, ntv, tid);          // dA = dM Yb^T
        if (s12a) mm_store<NT, NCOL, true>(acc, vOA, o_off, A, N, ldr, nv, ntv, sa1, sa2, tid);
        else mm_store<NT, NCOL, false>(acc, vOA, o_off, A, N, ldr, nv, ntv, sa1, sa2, tid);
        mm_gemm<NT, NCOL, false, false>(acc, A, D, mm_lds, ldr, nv, ntv, tid);        // dB = Ya^T dM
        if (s12a) mm_store<NT, NCOL, true>(acc, vOB, o_off, B, N, ldr, nv, ntv, sb1, sb2, tid);
        else mm_store<NT, NCOL, false>(acc, vOB, o_off, B, N, ldr, nv, ntv, sb1, sb2, tid);
    }
    if (s12a) {
        sa1 = wave_sum(sa1);
        sa2 = wave_sum(sa2);
        sb1 = wave_sum(sb1);
        sb2 = wave_sum(sb2);
        if (lane == 0) {
            red[wv][0] = sa1;
            red[wv][1] = sa2;
            red[wv][2] = sb1;
            red[wv][3] = sb2;
        }
        __syncthreads();
        if (tid < 4) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < MM_NW; ++w) v += red[w][tid];                      // fixed order
            float *dst = (tid < 2 ? s12a : s12b) + (long long)gc * 2 + (tid & 1);
            *dst = v;
        }
    }
}

template <int NT, int NCOL = NT>
int launch_fwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr, void *out,
                 long long ogstride, long long ldo, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)chan_matmul_fwd16_kernel<NT, NCOL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  MMCfg<NT>::LDS_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL((chan_matmul_fwd16_kernel<NT, NCOL>), dim3(G * ya->C), dim3(MM_THREADS), MMCfg<NT>::LDS_BYTES, st, *ya, *yb,
                       nvalid, N, ldr, G, out, ogstride, ldo);
    FGNN_LAUNCH_CHECK();
    return 0;
}
template <int NT, int NCOL = NT>
int launch_bwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmg, long long ldm, const int *nvalid,
                 int G, int N, int ldr, void *da, void *db, long long ogstride, long long ldo, float *s12a, float *s12b,
                 hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)chan_matmul_bwd16_kernel<NT, NCOL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  MMCfg<NT>::LDS_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL((chan_matmul_bwd16_kernel<NT, NCOL>), dim3(G * ya->C), dim3(MM_THREADS), MMCfg<NT>::LDS_BYTES, st, *ya, *yb, dm,
                       dmg, ldm, nvalid, N, ldr, G, da, db, ogstride, ldo, s12a, s12b);
    FGNN_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int fgnn_chan_matmul_fwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const int *nvalid, int G, int N, int ldr,
                                      void *out, long long ogstride, long long ldo, void *stream) {
    FGNN_CHECK(ya && yb && out && ya->ptr && yb->ptr, "fgnn_chan_matmul_fwd16: null argument");
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0 && N <= 256, "fgnn_chan_matmul_fwd16: bad shapes (N <= 256)");
    FGNN_CHECK(ldr >= N && ldr % 8 == 0, "fgnn_chan_matmul_fwd16: ldr must be a multiple of 8 and >= N");
    FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 2 && (long long)G * yb->gstride < 0x7fffffffll / 2 &&
               (long long)G * ogstride < 0x7fffffffll / 2,
               "fgnn_chan_matmul_fwd16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return launch_fwd16<2>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, st);
    if (N <= 128) return launch_fwd16<4>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, st);
    if (N <= 224) return launch_fwd16<8, 7>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, st);   // 7 tile columns: 16 registers less
    return launch_fwd16<8>(ya, yb, nvalid, G, N, ldr, out, ogstride, ldo, st);
}

extern "C" int fgnn_chan_matmul_bwd16(const fgnn_slab16 *ya, const fgnn_slab16 *yb, const void *dm, long long dmgstride,
                                      long long ldm, const int *nvalid, int G, int N, int ldr, void *da, void *db,
                                      long long ogstride, long long ldo, float *s12a, float *s12b, void *stream) {
    FGNN_CHECK(ya && yb && dm && da && db && ya->ptr && yb->ptr, "fgnn_chan_matmul_bwd16: null argument");
    FGNN_CHECK(ya->C == yb->C && ya->C > 0 && G > 0 && N > 0 && N <= 256, "fgnn_chan_matmul_bwd16: bad shapes (N <= 256)");
    FGNN_CHECK(ldr >= N && ldr % 8 == 0, "fgnn_chan_matmul_bwd16: ldr must be a multiple of 8 and >= N");
    FGNN_CHECK((s12a == nullptr) == (s12b == nullptr), "fgnn_chan_matmul_bwd16: s12a and s12b come together");
    FGNN_CHECK(!s12a || (ya->nrm && yb->nrm), "fgnn_chan_matmul_bwd16: s12 outputs need normalised slabs");
    FGNN_CHECK((long long)G * ya->gstride < 0x7fffffffll / 2 && (long long)G * yb->gstride < 0x7fffffffll / 2 &&
               (long long)G * dmgstride < 0x7fffffffll / 2 && (long long)G * ogstride < 0x7fffffffll / 2,
               "fgnn_chan_matmul_bwd16: a tensor exceeds 2 GiB (32-bit buffer addressing); split the batch");
    hipStream_t st = (hipStream_t)stream;
    if (N <= 64) return launch_bwd16<2>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, st);
    if (N <= 128) return launch_bwd16<4>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, st);
    if (N <= 224) return launch_bwd16<8, 7>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, st);
    return launch_bwd16<8>(ya, yb, dm, dmgstride, ldm, nvalid, G, N, ldr, da, db, ogstride, ldo, s12a, s12b, st);
}
