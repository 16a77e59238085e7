// fp32 channel contractions on the bf16 matrix cores: the "x3" operand split (gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate and shares its issue slots with the fp32 VALU work of the
// same SIMD (MI355X_MICROARCH.md: 64 cycles per instruction; measured in tools/ubench_mfma.hip).  The x3 kernels keep
// every tensor fp32 in HBM and in registers, but feed the contraction to v_mfma_f32_32x32x16_bf16: each fp32 operand
// value is written EXACTLY as the sum of three bf16 numbers
//     x = x0 + x1 + x2,   x0 = rn_bf16(x), x1 = rn_bf16(x - x0), x2 = x - x0 - x1      (8 + 8 + 8 significant bits)
// and a product of two operands is accumulated in fp32 from the six partial products of weight >= 2^-16,
//     a b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0);
// the three dropped ones are below 2^-24 |a b| each, i.e. below the rounding error of one fp32 fma.  Every bf16 x bf16
// product is exact in fp32 and the accumulation is fp32, so the result differs from the fp32-MFMA chain by the usual
// reassociation noise of an fp32 sum (the parity gates of tests/test_gpu_parity.py are the same for both kernel sets).
// Six bf16 MFMAs of K = 16 cost 6 x 32 = 192 cycles where eight fp32 MFMAs of K = 2 cost 512, and -- unlike those --
// they run beside the VALU work (the split itself: 5.5 instructions per operand value).
#pragma once
#include "fgnn_bf16.h"

// Partial products per operand pair (template parameter TERMS of the GEMM helpers): 6 = all of weight >= 2^-16 (dropped:
// a1 b2, a2 b1 at 2^-24 and a2 b2 at 2^-32), 8 = those two as well (products exact to 2^-32: what is left is the fp32
// accumulation).  The forward chain -- whose outputs decide ReLU masks and arg-max indices, in the forward kernels AND in the
// backward's recompute, which must reproduce them bit for bit -- runs with X3_FWD_TERMS, gradient GEMMs with X3_BWD_TERMS.
#ifndef X3_FWD_TERMS
#define X3_FWD_TERMS 8
#endif
#ifndef X3_BWD_TERMS
#define X3_BWD_TERMS 6
#endif

// ---- operand images: three bf16 part-images [part][step][lane][4 dwords] + the fp32 bias tail [layer][h][16] ------------
// The step sequence of an image is the bf16 kernels' (pk16_layout / pk16_value in fgnn_bf16.h).
struct PkX3 {
    Pk16 p;
    int part_dw, bias_off, floats;
};
HD16 constexpr PkX3 pkx3_layout(int kind, int ca, int cb, int depth) {
    PkX3 x{};
    x.p = pk16_layout(kind, ca, cb, depth);
    x.part_dw = x.p.steps * 256;
    x.bias_off = 3 * x.part_dw;
    x.floats = x.bias_off + 32 * x.p.nbias;
    return x;
}

// three bf16 parts of 16 fp32 values held in D-fragment order (value r <-> channel / pixel ch_of(r, h)): part q as a packed
// fragment (dword d = values 2d, 2d + 1), i.e. the operand layout of mfma16 (fgnn_bf16.h)
struct X3 {
    F16 p[3];
};
DEVI void split_pair(float a, float b, unsigned &d0, unsigned &d1, unsigned &d2) {
    d0 = cvt_pk(a, b);
    const float ra = a - bf_lo(d0), rb = b - bf_hi(d0);         // exact
    d1 = cvt_pk(ra, rb);
    const float la = ra - bf_lo(d1), lb = rb - bf_hi(d1);       // exact, <= 8 significant bits left
    d2 = cvt_pk(la, lb);
}
DEVI void split16(X3 &x, const float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) split_pair(v[2 * q], v[2 * q + 1], x.p[0].d[q], x.p[1].d[q], x.p[2].d[q]);
}
DEVI void split16(X3 &x, const f32x16 &v) {
#pragma unroll
    for (int q = 0; q < 8; ++q) split_pair(v[2 * q], v[2 * q + 1], x.p[0].d[q], x.p[1].d[q], x.p[2].d[q]);
}

// The same split with the two residual subtractions on the (otherwise mostly idle) matrix pipe: with the D-fragment values
// v as accumulator input and A = -I (in the k-slot order of the fragments), D = v - x0 EXACTLY (one non-zero product per
// output, the difference is representable), so a 16-value fragment costs 24 conversions + 4 MFMAs instead of 88 VALU
// instructions.  `negI` = make_neg_identity(lane), kept in registers across the tile loop.
DEVI F16 make_neg_identity(int lane) {
    const int n = lane & 31, h = lane >> 5;
    F16 f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int t = q >> 2, s0 = 2 * (q & 3);
        const unsigned lo = ch_of(8 * t + s0, h) == n ? 0xBF80u : 0u;             // bf16(-1)
        const unsigned hi = ch_of(8 * t + s0 + 1, h) == n ? 0xBF800000u : 0u;
        f.d[q] = lo | hi;
    }
    return f;
}
DEVI void split16m(X3 &x, const f32x16 &v, const F16 &negI) {
    pack_acc(x.p[0], v);
    f32x16 r = mfma16(step_of(negI, 0), step_of(x.p[0], 0), v);
    r = mfma16(step_of(negI, 1), step_of(x.p[0], 1), r);
    pack_acc(x.p[1], r);
    r = mfma16(step_of(negI, 0), step_of(x.p[1], 0), r);
    r = mfma16(step_of(negI, 1), step_of(x.p[1], 1), r);
    pack_acc(x.p[2], r);
}
DEVI void split16m(X3 &x, const float (&v)[16], const F16 &negI) {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = v[r];
    split16m(x, a, negI);
}

// part q, step s of an image in LDS
DEVI i32x4 x3_step(const float *wl, int part_dw, int q, int step, int lane) {
    return reinterpret_cast<const i32x4 *>(wl + q * part_dw)[step * 64 + lane];
}

// acc += W(steps s0, s0+1 ...) * X over NSTEP k-steps of 16: the six products, smallest first.
// W is the A operand (lane = output row), X the B operand (lane = column).
template <int NSTEP, int TERMS>
DEVI f32x16 gemm_x3(f32x16 acc, const float *wl, int part_dw, int s0, const X3 &x, int lane) {
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
        const i32x4 w0 = x3_step(wl, part_dw, 0, s0 + t, lane), w1 = x3_step(wl, part_dw, 1, s0 + t, lane),
                    w2 = x3_step(wl, part_dw, 2, s0 + t, lane);
        const i32x4 x0 = step_of(x.p[0], t), x1 = step_of(x.p[1], t), x2 = step_of(x.p[2], t);
        if constexpr (TERMS >= 8) {
            acc = mfma16(w1, x2, acc);
            acc = mfma16(w2, x1, acc);
        }
        acc = mfma16(w0, x2, acc);
        acc = mfma16(w2, x0, acc);
        acc = mfma16(w1, x1, acc);
        acc = mfma16(w0, x1, acc);
        acc = mfma16(w1, x0, acc);
        acc = mfma16(w0, x0, acc);
    }
    return acc;
}
// both operands in registers (weight-gradient GEMMs: A = d(pre-activation), B = layer input, contraction over pixels)
template <int TERMS>
DEVI f32x16 gemm_x3_rr(f32x16 acc, const X3 &a, const X3 &b) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const i32x4 a0 = step_of(a.p[0], t), a1 = step_of(a.p[1], t), a2 = step_of(a.p[2], t);
        const i32x4 b0 = step_of(b.p[0], t), b1 = step_of(b.p[1], t), b2 = step_of(b.p[2], t);
        if constexpr (TERMS >= 8) {
            acc = mfma16(a1, b2, acc);
            acc = mfma16(a2, b1, acc);
        }
        acc = mfma16(a0, b2, acc);
        acc = mfma16(a2, b0, acc);
        acc = mfma16(a1, b1, acc);
        acc = mfma16(a0, b1, acc);
        acc = mfma16(a1, b0, acc);
        acc = mfma16(a0, b0, acc);
    }
    return acc;
}

// Transposed operand of a weight-gradient GEMM (contraction over the tile's 32 pixels) WITHOUT LDS: each bf16 part of a
// normal-form operand (lane = pixel, slots = channels) goes through the matrix pipe against -I (two MFMAs, exact: every
// product is x * -1 or x * 0), the fp32 result (lane = channel, register r <-> pixel ch_of(r, h)) is re-packed (exact: the
// values are bf16 numbers).  The result is the NEGATED transposed operand; a GEMM of two such operands is the product of the
// un-negated ones.  WITH_SUM: nsum += (-1) x the sum of the lane's 16 pixels over the three parts = the lane's share of a
// bias gradient (rows of the operand summed over the pixels), in a fixed order.
template <bool WITH_SUM>
DEVI void transpose_x3(X3 &t, const X3 &x, const F16 &negI, float &nsum) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ts = 0.f;                                        // this tile's sum: parts smallest first, then into the running sum
#pragma unroll
    for (int q = 2; q >= 0; --q) {
        f32x16 a = mfma16(step_of(x.p[q], 0), step_of(negI, 0), zero);
        a = mfma16(step_of(x.p[q], 1), step_of(negI, 1), a);
        if constexpr (WITH_SUM) {
            const float s0 = (a[0] + a[1]) + (a[2] + a[3]), s1 = (a[4] + a[5]) + (a[6] + a[7]);
            const float s2 = (a[8] + a[9]) + (a[10] + a[11]), s3 = (a[12] + a[13]) + (a[14] + a[15]);
            ts += (s0 + s1) + (s2 + s3);
        }
        pack_acc(t.p[q], a);
    }
    if constexpr (WITH_SUM) nsum += ts;
}

// Weight-gradient GEMM with the second operand streamed: acc += A (x) B over the tile's pixels, A = an already transposed
// operand (three parts), B = the fp32 values v in normal form (lane = pixel, slots = channels).  Part q of B is split off, sent
// through the matrix-pipe transposer and consumed by its products at once (b0: a2 b0, a1 b0, a0 b0; b1: a1 b1, a0 b1; b2: a0 b2
// -- the six products of weight >= 2^-16), so beside A only ONE 8-register part of B exists at a time, in either form.
DEVI f32x16 wgrad_stream_x3(f32x16 acc, const X3 &a, const f32x16 &v, const F16 &negI) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 r = v;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        F16 p;
        pack_acc(p, r);
        if (q < 2) {                                        // residual for the next part: r - p, exact
            r = mfma16(step_of(negI, 0), step_of(p, 0), r);
            r = mfma16(step_of(negI, 1), step_of(p, 1), r);
        }
        f32x16 t = mfma16(step_of(p, 0), step_of(negI, 0), zero);
        t = mfma16(step_of(p, 1), step_of(negI, 1), t);
        F16 b;
        pack_acc(b, t);                                     // (negated, like A: the product of the two is not)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const i32x4 bs = step_of(b, s);
            if (q == 0) acc = mfma16(step_of(a.p[2], s), bs, acc);
            if (q <= 1) acc = mfma16(step_of(a.p[1], s), bs, acc);
            acc = mfma16(step_of(a.p[0], s), bs, acc);
        }
    }
    return acc;
}
DEVI f32x16 wgrad_stream_x3(f32x16 acc, const X3 &a, const float (&v)[16], const F16 &negI) {
    f32x16 x;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = v[r];
    return wgrad_stream_x3(acc, a, x, negI);
}
